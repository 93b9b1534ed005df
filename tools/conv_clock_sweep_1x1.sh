mkdir -p gpurun_out/r3g
run() { timeout -k 10 100 python tools/conv_clock.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v "per WG"; }
{
echo "== 1x1 16x16 256->384"; run auto 16 256 384 1x1; run 1,1,2 16 256 384 1x1; run 1,1,4 16 256 384 1x1; run 0,1,2 16 256 384 1x1; run 2,1,4 16 256 384 1x1
echo "== 1x1 16x16 128->384"; run auto 16 128 384 1x1; run 1,1,2 16 128 384 1x1; run 0,1,2 16 128 384 1x1
echo "== 1x1 8x8 256->384"; run auto 8 256 384 1x1; run 1,1,4 8 256 384 1x1; run 2,1,2 8 256 384 1x1
echo "== 1x1 16x16 512->128 (res conv of the up path)"; run auto 16 512 128 1x1; run 1,1,4 16 512 128 1x1; run 2,2,4 16 512 128 1x1
} > gpurun_out/r3g/conv_clock3.txt 2>&1
cat gpurun_out/r3g/conv_clock3.txt
