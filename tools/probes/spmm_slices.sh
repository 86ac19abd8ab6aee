cd $GRAFT_REPO_ROOT
for l in 1 2 4; do
  KGCN_HIP_LIB=$PWD/build/variants/libkgcn_dev.so KGCN_SPMM_SLICES=$l timeout 300 python tools/probes/spmm_slice_shapes.py 2>/dev/null > gpurun_out/slices_$l.json
done
python - <<'P'
import json
r={l: json.load(open("gpurun_out/slices_%d.json" % l)) for l in (1,2,4)}
for k in r[1]:
    print("%-24s" % k, "  ".join("sl>=%d: %7.1f us %.3f" % (l, r[l][k][0], r[l][k][1]) for l in (1,2,4)))
P
