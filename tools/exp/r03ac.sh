#!/bin/bash
for rep in 1 2 3; do
  echo "== HEAD"
  (cd tools/exp/headcopy && python tools/train_profile.py --batch 8 --plain 2>&1 | grep "^batch")
  echo "== current"
  python tools/train_profile.py --batch 8 --plain 2>&1 | grep "^batch"
done
