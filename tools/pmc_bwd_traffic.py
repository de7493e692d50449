"""Reduce the FETCH_SIZE / WRITE_SIZE rows of the DCNv2 backward PMC summaries (tools/run_pmc_bwd.sh with PMC_WHAT=data and
PMC_WHAT=weight; tools/pmc_dcn_bwd.py: 64->64 @256x512 x4, the launch that dominates the B = 4 training step) into
profiles/dcn_bwd_pmc.json, which bench.py reads for train.roofline_bwd_*.traffic -- trusted only at the kernel revision
(source hash) it was taken on.  Usage: python3 tools/pmc_bwd_traffic.py <data summary> <weight summary> <out json>"""
import json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench


def parse(path):
    out, cur = {}, None
    for line in open(path):
        m = re.match(r"== (\S+?)[<( ]", line + " ")
        if line.startswith("== "):
            cur = line[3:].split("  launches")[0].strip()
            out[cur] = {}
            continue
        m = re.match(r"\s+(\w+)\s+avg/launch\s+([0-9.e+]+)", line)
        if m and cur:
            out[cur][m.group(1)] = float(m.group(2))
    return out


data, weight = parse(sys.argv[1]), parse(sys.argv[2])
traffic = lambda c: (2.0 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024.0
dk = [k for k in data if k.startswith("dcn_bwd_data2_kernel") or k.startswith("dcn_bwd_gx_reduce_kernel")]
wk = [k for k in weight if k.startswith("dcn_bwd_weight")]
out = {"kernel_rev": bench.kernel_revision(bench.DCN_BWD_SOURCES),
       "inputs": "tools/pmc_dcn_bwd.py (unit-normal x / grad_out, 0.5-px offsets), 6 launches per pass",
       "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE | WRITE_SIZE --output-format csv -- python3 tools/pmc_dcn_bwd.py (separate passes, tools/run_pmc_bwd.sh)",
       "correction": "traffic = (2*FETCH_SIZE + WRITE_SIZE)*1024 bytes summed over the family's kernels (gfx950: FETCH_SIZE reports "
                     "half of a wide coalesced stream's bytes; these kernels stage rows with dword / 16-byte loads, for which the "
                     "doubling is an upper bound)",
       "layers": {"4x64x64x256x512": {"dcn_bwd_data": sum(traffic(data[k]) for k in dk),
                                      "dcn_bwd_weight": sum(traffic(weight[k]) for k in wk)}},
       "_raw": {"dcn_bwd_data": {k: {c: data[k][c] for c in ("FETCH_SIZE", "WRITE_SIZE")} for k in dk},
                "dcn_bwd_weight": {k: {c: weight[k][c] for c in ("FETCH_SIZE", "WRITE_SIZE")} for k in wk}}}
json.dump(out, open(sys.argv[3], "w"))
print(json.dumps(out)[:600])
