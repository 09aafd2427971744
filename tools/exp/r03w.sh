#!/bin/bash
for extra in "" "--fused-adam" "--blas cublas" "--blas cublaslt" "--fused-adam --blas cublas"; do
  echo "== $extra"
  python tools/train_profile.py --batch 8 --plain $extra 2>&1 | grep "^batch"
done
