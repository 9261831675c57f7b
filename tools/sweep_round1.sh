B="python bench.py --queries 16384 --steps 3 --warmup 1 --cpu-queries 0"
f() { "$@" 2>&1 | tail -1 | grep -o "\"value[^,]*\|kernel_ms[^,]*\|merge_ms[^,]*" | tr '\n' ' '; echo; }
for w in 8 12 16 24; do echo prunemin=$w div=2048; UGP_COARSE_DIV=2048 UGP_PRUNE_MIN_WORDS=$w f $B; done
for d in 4096 16384; do echo prunemin=24 coarsediv=$d; UGP_PRUNE_MIN_WORDS=24 UGP_COARSE_DIV=$d f $B; done
