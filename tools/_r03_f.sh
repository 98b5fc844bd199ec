set -e
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout -k 10 420 python bench.py > gpurun_out/r03f_bench_line.json 2> gpurun_out/r03f_bench.err || { tail -30 gpurun_out/r03f_bench.err; exit 1; }
tail -4 gpurun_out/r03f_bench.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r03f_bench_line.json"))
print({k:d[k] for k in ("value","ms_per_step")})
print("roofline", d["roofline"]["achieved"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["traffic_source"])
rs=d["roofline_set"]
print({k:rs[k] for k in rs if k.startswith("set_") and ("frac" in k or "ms" in k)})
for p in rs["per_size"]: print(p)
for k,v in (d.get("configs") or {}).items(): print(k, {a:v.get(a) for a in ("act_capi_ms","act_product_ms","act_product_graph_ms","weight_capi_ms","weight_product_ms","set_capi_GBps","set_product_GBps","error")})
print(d["cpu_baseline"])
PY
timeout -k 10 600 bash tools/profile_bench.sh r03 > gpurun_out/r03f_profile.log 2>&1 || { tail -20 gpurun_out/r03f_profile.log; exit 1; }
tail -5 gpurun_out/r03f_profile.log
ls gpurun_out/prof_r03
