mkdir -p gpurun_out
{
for v in d f; do echo "== full $v";  ./scripts/probe/igemm_trace_$v 0 8 16 128 256 256; done
} > gpurun_out/abl_mid3.txt 2>&1
