#!/bin/bash
# per-workload work statistics of the tracking kernel: lookups, loads, trips, lanes (args: "workload spp" ...)
for w in "$@"; do
  set -- $w
  timeout -k 10 300 python -u bench.py --workload $1 --spp $2 --steps 2 --warmup 1 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); dv=d['divergence']; r=d['roofline']
spl=d['config']['image'].split('x'); samples=int(spl[0])*int(spl[1])*d['config']['spp']
print('$1 spp$2', 'Msamples/s', round(d['value'],1), 'kernel_ms', round(r['kernel_ms_per_launch'],2), 'lookups/sample', round(d['lookups_per_sample'],1), 'loads/sample', round(r['loads']['loads_per_lookup']*d['lookups_per_sample'],2), 'trips/sample(x64)', round(dv['wave_trips']*64/samples,2), 'lanes/trip', round(dv['walking_lanes_per_trip'],1), 'phases/sample(x64)', round(dv['event_phases']*64/samples,3), 'scatter/sample', round(dv['scatter_events']/samples,3), 'loads GB/s', round(r['loads']['achieved'],1))"
done
