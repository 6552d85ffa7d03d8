"""Fold rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (scripts/collect_profiles.sh) into
profiles/pmc_summary.json: per kernel class, the average HBM bytes per launch, corrected as
/opt/skills/guides/MI355X_MICROARCH.md prescribes for gfx950: bytes = (2*FETCH_SIZE + WRITE_SIZE) * 1024
(FETCH_SIZE counts half of the wide reads; both counters are in KiB). The x2 is checked on a kernel whose
traffic is known exactly: k_gen_probes writes one panel sweep, the ||v||^2 sweep k_axpy_norm<.,.,1> reads one."""

import csv
import glob
import hashlib
import json
import os
import re
import sys
from collections import defaultdict
from pathlib import Path

out_dir = sys.argv[1]
ROOT = Path(__file__).resolve().parent.parent


def kernel_sources_sha256():
	"""Same digest as bench.py:kernel_sources_sha256 - the bench line quotes a traffic figure only when it matches."""
	h = hashlib.sha256()
	for f in ("primate_amd/csrc/slq_common.hpp", "primate_amd/csrc/slq_kernels.hpp", "primate_amd/csrc/slq_ring.hpp", "primate_amd/csrc/slq_ring_fa.hpp", "primate_amd/csrc/slq_build.hpp", "primate_amd/csrc/slq.hip"):
		h.update((ROOT / f).read_bytes())
	return h.hexdigest()



def classify(name: str, orth: int):
	m = re.search(r"k_csr_pass<(\w+), (\d+), (\d), ", name)
	if m:
		return {"0": "spmm_3term", "1": "reorth_dot", "3": "reorth_dot", "2": "reorth_update" if orth > 0 else "axpy_norm"}[m.group(3)]
	m = re.search(r"k_csr_(?:ring|tile)_pass<(\w+), (\d), ", name)  # the tiled forms of the same passes: <F, PASS, ...>
	if m:
		return {"0": "spmm_3term", "4": "spmm_3term", "1": "reorth_dot", "3": "reorth_dot", "2": "reorth_update" if orth > 0 else "axpy_norm"}[m.group(2)]
	m = re.search(r"k_ring_pass<(\w+), (\d), ", name)  # slq_ring.hpp: <F, PASS, NTP, RC, LPR, WAVES>; PASS 5 = update pass with Gram rows
	if m:
		return {"0": "spmm_3term", "4": "spmm_3term", "3": "reorth_dot", "2": "reorth_update" if orth > 0 else "axpy_norm", "5": "reorth_update"}[m.group(2)]
	for k, c in (("k_spmm_3term", "spmm_3term"), ("k_reorth_dot", "reorth_dot"), ("k_reorth_update", "reorth_update")):
		if k in name:
			return c
	if "k_axpy_norm" in name:
		return "probe_norm" if re.search(r"k_axpy_norm<\w+, \d+, 1>", name) else "axpy_norm"
	if "k_gen_probes" in name:
		return "gen_probes"
	return None


summary = {"_meta": {"tag": os.path.basename(os.path.normpath(out_dir)), "kernel_sha256": kernel_sources_sha256(),
                     "collected_by": "scripts/collect_profiles.sh (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, one counter per pass)"}}
for workload, orth, P in (("lap2d_1000", 3, 256), ("lap2d_1000", 0, 256), ("lap2d_1000", 6, 256), ("lap2d_1000", 30, 256), ("lap3d_100", 3, 256), ("lap3d_100", 0, 256),
                          ("lap3d_100", 30, 256), ("lap2d_1000", 3, 64), ("lap3d_100", 3, 64)):
	per = defaultdict(lambda: defaultdict(list))
	names = {}
	sfx = "" if P == 256 else f"_p{P}"
	for counter in ("FETCH_SIZE", "WRITE_SIZE"):
		for f in glob.glob(f"{out_dir}/pmc_{counter}_{workload}_orth{orth}{sfx}/**/*counter_collection.csv", recursive=True):
			with open(f) as fh:
				for row in csv.DictReader(fh):
					if row.get("Counter_Name") != counter:
						continue
					cls = classify(row["Kernel_Name"], orth)
					if cls:
						per[cls][counter].append(float(row["Counter_Value"]))
						names[cls] = row["Kernel_Name"].split("(")[0]
	if not per:
		continue
	entry = {}
	for cls, d in per.items():
		fs, ws = d.get("FETCH_SIZE", []), d.get("WRITE_SIZE", [])
		if not fs or not ws:
			continue
		fa, wa = sum(fs) / len(fs), sum(ws) / len(ws)
		entry[cls] = {
			"kernel": names[cls],
			"launches_sampled": min(len(fs), len(ws)),
			"FETCH_SIZE_KiB_avg": fa,
			"WRITE_SIZE_KiB_avg": wa,
			"hbm_bytes_per_launch": int((2 * fa + wa) * 1024),
			"correction": "(2*FETCH_SIZE + WRITE_SIZE)*1024 per MI355X_MICROARCH.md HBM section; check rows: gen_probes writes and probe_norm reads exactly one panel sweep (s*n*b bytes)",
		}
	summary[f"{workload}/P{P}/k30/orth{orth}"] = entry
json.dump(summary, sys.stdout, indent=1)
