#!/bin/bash
# developer diagnostic: how often the per-trip empty-space service and voxel lookup run (needs tools/alt/libvr_diag.so)
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so; cp tools/alt/libvr_diag.so volume_renderer_amd/csrc/libvr_backend.so
python -u bench.py --spp 16 --steps 1 --warmup 0 --cpu-baseline off 2>/dev/null | tail -1 | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); v=d['divergence']; print('trips', v['wave_trips'], 'with empty service', v['event_phases'], round(v['event_phases']/v['wave_trips'],3), 'with lookup', v['wave_trip_lanes'], round(v['wave_trip_lanes']/v['wave_trips'],3))"
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
