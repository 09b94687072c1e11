# forward block-split threshold sweep (analysis):  gpurun -- bash tools/sweep_split.sh
for m in 100000 512 384 256 192 128; do
  echo "split_min $m"; GFL_FWD_SPLIT_MIN=$m python bench.py --steps 200 --warmup 50 --no-clip --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms']['blend_fwd'])"
  GFL_FWD_SPLIT_MIN=$m bash tools/clip_kernels.sh 3 10 2>&1 | grep "blend_fwd_kernel"
done
