#!/bin/bash
# development: build the product library, the measurement build and variant libraries in parallel
#   scripts/build_variants.sh "tag:DEF=1,DEF2=3" "tag2:..."
cd "$(dirname "$0")/.."
python -c "from neural_renderer_amd import _build; _build.build(force=True)" &
python -c "from neural_renderer_amd import _build; _build.build_profile(force=True)" &
for v in "$@"; do
  tag=${v%%:*}; defs=${v#*:}
  python -c "from neural_renderer_amd import _build; _build.build_variant('$tag', '$defs'.split(','))" &
done
wait
ls -la neural_renderer_amd/*.so
