# A/B of whole bench lines (GPU box): AB_VARIANTS="a b base" bash tools/ab_bench.sh — variants are source_amd/lib/variants/librsx_<name>.so (e.g. built from an earlier commit in a git worktree); prints ms/step, trace ms, accumulate ms
# NOTE: bench.py rebuilds the in-tree librsx.so on the box when a source is newer than it — "base" is then the CURRENT sources, not the library you built before the edit: compare variant libraries (RSX_LIB) with each other.
R=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2 3 4; do for v in ${AB_VARIANTS:-base}; do lib=$R/source_amd/lib/variants/librsx_$v.so; [ "$v" = base ] && lib=$R/source_amd/lib/librsx.so
for w in c3; do echo -n "$v $w "; RSX_LIB=$lib python $R/bench.py --workload $w --no-pmc --no-cpu-baseline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['accumulate_kernel_ms'])"; done; done; done
