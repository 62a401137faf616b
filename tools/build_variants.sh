#!/bin/bash
# developer tool: build several libpomgpu variants in parallel: tools/build_variants.sh "tag1:-DA=1 -DB=2" "tag2:-DC" ...
ROOT=$(cd "$(dirname "$0")/.." && pwd)
for spec in "$@"; do
  tag=${spec%%:*}; flags=${spec#*:}
  $ROOT/tools/build_variant.sh $tag $flags > /tmp/build_$tag.log 2>&1 &
done
wait
ls -la $ROOT/build_variants/
