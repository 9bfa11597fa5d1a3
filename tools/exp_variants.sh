for v in "512 32" "1024 16" "2048 8" "4096 4"; do
  set -- $v
  python -c "from katsdpimager_amd import build; build.build_lib(force=True, extra_flags=['-DKIMG_INTERLEAVE_MIN_CHUNK=$1', '-DKIMG_INTERLEAVE_MAX_PARTS=$2'])" > gpurun_out/build_v.log 2>&1
  echo "== min_chunk $1 max_parts $2"
  timeout -k 10 300 python bench.py --no-secondary --cpu-sample 0 --steps 5 2>/dev/null > gpurun_out/bench_v.json; python tools/show_bench.py gpurun_out/bench_v.json | head -2
done
