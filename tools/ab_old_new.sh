#!/bin/bash
# same-box A/B of the snapshot paths: the tests first, then 8-frame clip fits with the snapshots on the side stream / in the
# snapshot iteration's own forward, alternating
python -m pytest tests/test_gpu_fused.py tests/test_gpu_fitvideo.py tests/test_gpu_fullsize.py -x -q -k "snapshot" 2>&1 | tail -5
for r in 1 2 3; do
  echo -n "[async] "; python tools/clip_repeat.py 5 8 async 2>&1 | tail -1 | cut -c1-150
  echo -n "[sync ] "; python tools/clip_repeat.py 5 8 sync 2>&1 | tail -1 | cut -c1-150
done
