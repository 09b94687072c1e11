for r in 1 2; do for m in 256 192 320 160; do
  echo -n "split $m: clip "; GFL_FWD_SPLIT_MIN=$m python tools/profile_clip.py 8 10 | grep "^total" | cut -d= -f2
  echo -n "split $m: step "; GFL_FWD_SPLIT_MIN=$m python bench.py --steps 200 --warmup 50 --no-clip --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['stage_ms']['blend_fwd'])"
done; done
