mkdir -p gpurun_out/peel
python tools/dev/ab_step.py 65536 0x40000,0x50000 prod peel2 prod peel2 > gpurun_out/peel/ab3.log 2>&1; grep -v amdgpu gpurun_out/peel/ab3.log | tail -20
