# same-box A/B of investigation builds of the backward: tools/ab_bwd.sh "<tag>[:ENV=VAL ...]" ...
D=$(ls -d scanerf-*/lib/debug)
for rep in 1 2; do
for spec in "$@"; do
  tag=${spec%%:*}; envs=""; [ "$spec" != "$tag" ] && envs=$(echo "${spec#*:}" | tr ':' ' ')
  echo -n "$spec: "; env $envs SCANERF_LIB=$D/libscanerf_hip_$tag.so timeout -k 10 100 python tools/bwd_emit_only.py 2>&1 | tail -1
done; done
