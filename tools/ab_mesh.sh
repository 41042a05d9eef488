#!/bin/bash
# tools/ab.sh over the mesh workloads: args = alternative libraries
for w in "config5_marcher_sphere_4k 16" "config5_marcher_nee_sphere_4k 16" "perlin_marcher_sphere_1080p 16" "config1_marcher_cube_512_1spp 1"; do
  W="$w" ./tools/ab.sh "$@" || exit 1
done
