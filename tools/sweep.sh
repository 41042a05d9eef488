#!/bin/bash
# developer sweep: work order (0 slot order, 1 rows centre-out, 2 pixel-major centre-out) vs framing, small and large frames
for dist in 2.2 4.0; do for spp in 4 16; do for ord in 0 1 2; do
  VR_BENCH_CAMERA_DIST=$dist VR_TUNE_ORDER=$ord timeout -k 10 120 python -u bench.py --spp $spp --steps 3 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['divergence']; print('dist $dist spp $spp order $ord', round(d['value'],1), 'Msamples/s', round(d['ms_per_step'],2), 'ms; lookups/sample', round(d['lookups_per_sample'],1), 'lanes', round(v['walking_lanes_per_trip'],2))"
done; done; done
