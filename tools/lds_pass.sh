#!/bin/bash
# LDS utilisation / bank-conflict share of the split GEMMs alone (tools/gemm_sweep.py) and of the recurrent kernels
# (tools/rec_time.py): one rocprofv3 --pmc pass each, kernel trace only.  -> gpurun_out/lds_{gemm,rec}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && cd $R
rm -rf gpurun_out/lds_gemm gpurun_out/lds_rec
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/lds_gemm -- python3 tools/gemm_sweep.py > gpurun_out/lds_gemm.log 2>&1 &&
REC_TIME_ITERS=8 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE --output-format csv -d gpurun_out/lds_rec -- python3 tools/rec_time.py > gpurun_out/lds_rec.log 2>&1
echo rc=$?
python3 - <<'PY'
import csv, glob, collections, re
for tag in ("gemm", "rec"):
    f = glob.glob(f"gpurun_out/lds_{tag}/*/*counter_collection.csv")
    if not f:
        print(tag, "no counters"); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"== {tag}: kernel | launches | LDS busy % of (GUI_ACTIVE/8 x 256 CUs) | bank-conflict cycles / LDS busy cycles")
    for k, d in acc.items():
        if not any(s in k for s in ("gemm_spike_kernel", "rec_fwd_kernel", "rec_bwd_kernel")):
            continue
        n = len(d["GRBM_GUI_ACTIVE"])
        act, conf, gui = (sum(d[c]) / n for c in ("SQ_LDS_IDX_ACTIVE", "SQ_LDS_BANK_CONFLICT", "GRBM_GUI_ACTIVE"))
        if gui <= 0 or act <= 0 or n < 3:
            continue
        name = re.sub(r"\(anonymous namespace\)::|^void ", "", k).split("(")[0][:80]
        print(f"{name} | {n} | {100 * act / (gui / 8 * 256):.1f} | {100 * conf / act:.1f} %")
PY
