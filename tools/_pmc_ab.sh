set -u
cd ${GRAFT_REPO_ROOT:-/root/repo}
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_ab; mkdir -p $OUT
for lib in nostream base; do
  if [ "$lib" = base ]; then unset RT_HIP_LIB; else export RT_HIP_LIB=$PWD/ray-tracing_amd/ab/librt_hip_$lib.so; fi
  timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_WAIT_INST_ANY --output-format csv -d $OUT/$lib -- python3 bench.py --scene random_scene --width 800 --height 450 --spp 1600 --steps 1 --warmup 0 --no-f32 --no-cpu-baseline > $OUT/$lib.json 2> $OUT/$lib.err || exit 1
done
python3 - <<'PY'
import csv,glob,collections
for lib in ("nostream","base"):
    tot=collections.defaultdict(float); n=0
    for f in glob.glob(f"gpurun_out/pmc_ab/{lib}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "render_kernel" in r["Kernel_Name"]:
                tot[r["Counter_Name"]]+=float(r["Counter_Value"])
    print(lib, {k:f"{v:.4g}" for k,v in sorted(tot.items())})
PY
