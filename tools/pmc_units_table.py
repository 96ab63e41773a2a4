#!/usr/bin/env python
"""Per-kernel unit utilisation from the four passes of tools/pmc_units.sh (<base>_p1..p4.csv, written by tools/summarize_pmc.py):
   python tools/pmc_units_table.py gpurun_out/units_l40
Counters are sums over a kernel's launches in the run; `active` = GRBM_GUI_ACTIVE / 8 (the counter is summed over the 8 XCDs) = the
launches' busy cycles.  Under --pmc the profiler serialises kernels: these are each kernel's figures with the chip to itself."""
import csv
import re
import sys

base = sys.argv[1]
passes = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
rows = {}
for i in passes:
    for r in csv.DictReader(open('%s_p%d.csv' % (base, i))):
        k = re.sub(r"\(.*", "", r['kernel']).replace('void ', '').replace('cliora::', '')[:44]
        d = rows.setdefault(k, {})
        for c, v in r.items():
            if c.endswith('_sum') and c != 'kernel':
                name = c[:-4]
                name = name[:-4] if name.endswith('_sum') else name
                d['%d.%s' % (i, name)] = float(v)
        d['n'] = int(r['dispatches'])
print('%-44s %5s %9s | %6s %8s | %8s %8s %7s %8s | %8s | %7s %7s %7s' % (
    'kernel', 'calls', 'cyc/call', 'LDS %', 'LDSconf%', 'L1 acc/c', 'L1->L2/c', 'L1 hit%', 'L1 pend%', 'L2 hit %', 'VALU %', 'LDSi %', 'MFMA %'))
for k, d in sorted(rows.items(), key=lambda kv: -kv[1].get('%d.GRBM_GUI_ACTIVE' % passes[0], 0))[:14]:
    def act(i):
        return d.get('%d.GRBM_GUI_ACTIVE' % i, 0) / 8.0
    def pct(key, i, units):
        return 100.0 * d.get('%d.%s' % (i, key), 0) / (act(i) * units) if act(i) else 0.0
    hit, miss = d.get('3.TCC_HIT', 0), d.get('3.TCC_MISS', 0)
    acc, rd = d.get('2.TCP_TOTAL_CACHE_ACCESSES', 0), d.get('2.TCP_TCC_READ_REQ', 0)
    a2 = act(2) * 256 or 1.0
    print('%-44s %5d %9.0f | %6.1f %8.1f | %8.3f %8.3f %7.1f %8.1f | %8.1f | %7.1f %7.1f %7.1f' % (
        k, d['n'], act(passes[0]) / d['n'], pct('SQ_LDS_IDX_ACTIVE', 1, 256), pct('SQ_LDS_BANK_CONFLICT', 1, 256),
        acc / a2, rd / a2, 100.0 * (1 - rd / acc) if acc else 0.0, 100.0 * d.get('2.TCP_PENDING_STALL_CYCLES', 0) / a2,
        100.0 * hit / (hit + miss) if hit + miss else 0.0,
        pct('SQ_ACTIVE_INST_VALU', 4, 1024) * 4, pct('SQ_ACTIVE_INST_LDS', 4, 1024) * 4, pct('SQ_VALU_MFMA_BUSY_CYCLES', 4, 1024)))
print("""
cyc/call   busy cycles per launch (GRBM_GUI_ACTIVE / 8 / launches)
LDS %      SQ_LDS_IDX_ACTIVE / (active x 256 CUs): share of cycles the CU's LDS pipe executes an instruction; LDSconf%: SQ_LDS_BANK_CONFLICT likewise
L1 acc/c   TCP_TOTAL_CACHE_ACCESSES per active cycle and CU (the vector L1 takes one tag lookup per cycle); L1->L2/c: TCP_TCC_READ_REQ likewise
           (a request = one 128-byte line: x 128 B x 2.4 GHz = the CU's ingest from L2 in bytes/s); L1 pend%: TCP_PENDING_STALL_CYCLES / (active x 256):
           share of cycles the L1 is stalled on returns it is waiting for
L2 hit %   TCC_HIT / (TCC_HIT + TCC_MISS)
VALU / LDSi %   SQ_ACTIVE_INST_* x 4 / (active x 1024 SIMDs) (waves issuing that class, summed over the SIMD's waves); MFMA %: SQ_VALU_MFMA_BUSY_CYCLES / (active x 1024)""")
