mkdir -p gpurun_out/r3g
run() { timeout -k 10 100 python tools/conv_clock.py "$@" 2>&1 | grep -v amdgpu.ids | grep -v timeline; }
{
echo "== s2 32x32 128->128"; run auto 32 128 128 s2; run 2,1,6 32 128 128 s2; run 2,1,4 32 128 128 s2; run 2,1,2 32 128 128 s2; run 4,1,2 32 128 128 s2; run 4,2,2 32 128 128 s2
echo "== s2 16x16 256->256"; run auto 16 256 256 s2; run 2,1,6 16 256 256 s2; run 4,1,2 16 256 256 s2
echo "== T 8x8 256->256"; run auto 8 256 256 T; run 1,1,4 8 256 256 T; run 1,1,2 8 256 256 T; run 2,1,6 8 256 256 T
echo "== T 16x16 128->128"; run auto 16 128 128 T; run 2,1,2 16 128 128 T; run 1,1,4 16 128 128 T
echo "== T 4x4 256->256"; run auto 4 256 256 T
} > gpurun_out/r3g/conv_clock2.txt 2>&1
cat gpurun_out/r3g/conv_clock2.txt
