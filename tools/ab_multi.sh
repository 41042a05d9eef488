#!/bin/bash
# tools/ab.sh over the four tracker regimes: args = alternative libraries
for w in "config2_ratio_1080p_64spp_cloud512 64" "config3_delta_1080p_256spp_cloud512 64" "config2_textbook 64" "config3_textbook 256"; do
  W="$w" ./tools/ab.sh "$@" || exit 1
done
