for a in 0 1 2 3 4 7 8 15; do
  echo "ablate=$a"; GPD_IMG_ABLATE=$a python bench.py --steps 5 --warmup 1 --cpu-samples 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels']['grasp_image_kernel']['ms'], d['kernels']['lenet_forward']['ms'])"
done
