#!/bin/bash
# A/B harness (GPU box): tools/ab.sh "<variants>" "<configs>" [steps] — runs tools/kbench.py for every variant library x config
# x LDS split in $AB_LDS ("world:mesh" pairs), un-pipelined unless AB_PIPE is set.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for cfg in $2; do
  for v in $1; do
    lib=$R/source_amd/lib/variants/librsx_$v.so; [ "$v" = base ] && lib=$R/source_amd/lib/librsx.so
    for lds in ${AB_LDS:-default}; do
      echo -n "$v lds=$lds pipe=${AB_PIPE:-1} "
      if [ "$lds" = default ]; then
        RSX_LIB=$lib RSX_PIPELINE=${AB_PIPE:-1} timeout 300 python $R/tools/kbench.py ${3:-3} $cfg | tail -1
      else
        RSX_LIB=$lib RSX_WORLD_LDS=${lds%%:*} RSX_MESH_LDS=${lds##*:} RSX_PIPELINE=${AB_PIPE:-1} timeout 300 python $R/tools/kbench.py ${3:-3} $cfg | tail -1
      fi
    done
  done
done
