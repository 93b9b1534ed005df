mkdir -p gpurun_out/r3g
run() { timeout -k 10 100 python tools/conv_clock.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "per WG"; }
{
echo "== s1 64x64 32->32 (B=32)"; run auto 64 32 32 s1; run 3,1,2 64 32 32 s1; run 4,1,2 64 32 32 s1
echo "== 1x1 64x64 64->32"; run auto 64 64 32 1x1; run 4,1,2 64 64 32 1x1; run 3,1,2 64 64 32 1x1
echo "== 1x1 64x64 32->64"; run auto 64 32 64 1x1; run 2,1,2 64 32 64 1x1; run 1,1,2 64 32 64 1x1
} > gpurun_out/r3g/conv_clock4.txt 2>&1
cat gpurun_out/r3g/conv_clock4.txt
