for v in ${SPLITS:-256 448 640 896 100000}; do
  echo "== split_min $v"
  GFL_FWD_SPLIT_MIN=$v python bench.py --no-clip --no-cpu-baseline --steps 100 --warmup 10 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); st=d['stage_ms']; print('  bench step %.4f ms  fwd %.1f us' % (d['ms_per_step'], 1e3*st['blend_fwd']))"
  GFL_FWD_SPLIT_MIN=$v python tools/clip_repeat.py 5 4 2>&1 | tail -1 | cut -c1-110
done
