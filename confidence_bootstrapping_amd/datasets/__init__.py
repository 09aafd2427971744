"""Data-side transforms of the fine-tuning loop (reference datasets/)."""
