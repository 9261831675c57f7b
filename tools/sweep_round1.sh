B="python bench.py --steps 3 --warmup 1 --cpu-queries 0"
f() { "$@" 2>&1 | tail -1 | grep -o "\"value[^,]*\|kernel_ms[^,]*\|merge_ms[^,]*" | tr '\n' ' '; echo; }
echo base; f $B
echo chunk610 unit16; UGP_CHUNK_NODES=610 UGP_UNIT_CHUNKS=16 f $B
echo chunk1220 unit8; UGP_CHUNK_NODES=1220 UGP_UNIT_CHUNKS=8 f $B
echo chunk305 unit32; UGP_CHUNK_NODES=305 UGP_UNIT_CHUNKS=32 f $B
