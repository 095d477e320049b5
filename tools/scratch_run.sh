cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3
python -m pytest tests/test_gpu_accumulate_image.py -x -q 2>&1 | tail -3
python tools/ab_syrk_image.py --n 4000000 --m 512 --reps 5 --ref 1 2>&1 | grep -v amdgpu.ids
python tools/ab_syrk_image.py --n 2500000 --m 1024 --reps 3 --ref 1 2>&1 | grep -v amdgpu.ids
