#!/bin/bash
for b in 2 4 8 16; do
  python tools/train_profile.py --batch $b --plain 2>&1 | grep "^batch"
done
