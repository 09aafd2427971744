#!/bin/bash
for rep in 1 2; do
for v in "1 0" "0 0" "1 1" "0 1"; do
  set -- $v
  echo "== side prio $1, record_stream $2"
  CBD_TRAIN_SIDE_PRIO=$1 CBD_TRAIN_RECORD_STREAM=$2 python tools/train_profile.py --batch 8 --plain 2>&1 | grep "^batch"
done
done
