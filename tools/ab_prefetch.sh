cd /root/repo
run() { # name so env...
  name=$1; so=$2; shift 2
  cp $so par_raytracer_amd/libprt_hip.so
  env "$@" python bench.py --no-cpu-baseline --no-other-workloads --pipeline 4 --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=j['roofline']
print('%-44s %8.1f Mrays/s %7.3f ms/frame kernel %7.3f ms util %s parity-rays %d' % ('$name', j['value'], j['ms_per_step'], r['kernel_ms_per_frame'], (r['lane_utilisation'] or {}).get('node_loop'), r['per_frame']['rays']))"
}
run "f0" variants/f0.so X=1
run "prefetch cap21" variants/prefetch.so PRT_STACK_CAP=21
run "prefetch cap21 keep48" variants/prefetch.so PRT_STACK_CAP=21 PRT_KEEP_MIN=48
run "prefetch cap21 keep56" variants/prefetch.so PRT_STACK_CAP=21 PRT_KEEP_MIN=56
run "f0 cap21" variants/f0.so PRT_STACK_CAP=21
run "f0 keep48" variants/f0.so PRT_KEEP_MIN=48
run "f0" variants/f0.so X=1
run "prefetch cap21" variants/prefetch.so PRT_STACK_CAP=21
cp variants/prefetch.so par_raytracer_amd/libprt_hip.so
PRT_STACK_CAP=21 python -m pytest tests -m gpu -q -x -k "golden or coincident or slow_paths" 2>&1 | tail -n 3
