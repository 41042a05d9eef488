#!/bin/bash
# local-majorant mode on each library given as argument (the tree's build first)
cp volume_renderer_amd/csrc/libvr_backend.so /tmp/base.so
for lib in /tmp/base.so "$@"; do
  cp $lib volume_renderer_amd/csrc/libvr_backend.so; echo "== $lib"
  bash tools/ab_local.sh || exit 1
done
cp /tmp/base.so volume_renderer_amd/csrc/libvr_backend.so
