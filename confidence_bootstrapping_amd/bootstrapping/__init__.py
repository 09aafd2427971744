"""Confidence-bootstrapping loop pieces (reference bootstrapping/)."""
