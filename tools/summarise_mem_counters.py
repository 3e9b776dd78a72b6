#!/usr/bin/env python3
"""Derived quantities from tools/pmc_mem_counters.sh's per-dispatch medians (one counter per rocprofv3 --pmc pass).

    python3 tools/summarise_mem_counters.py gpurun_out/pmc_mem_cfg3.txt > profiles/r04/cfg3_mem_counters.txt

Instance counts on gfx950: 8 XCDs x 16 L2 channels = 128 TCC instances, 256 TCP / TA instances (one per CU),
GRBM_GUI_ACTIVE summed over the 8 XCDs.  EA read requests are tallied per 128-byte line on this part, EA write
requests per 64 bytes (MI355X_MICROARCH.md, HBM section) — the byte figures below reproduce FETCH_SIZE / WRITE_SIZE."""
import re, sys
txt = open(sys.argv[1]).read()
kern, cur = {}, None
for line in txt.splitlines():
    m = re.match(r"^(\w+)$", line.strip())
    if m and not line.startswith(" "):
        cur = kern.setdefault(m.group(1), {})
        continue
    m = re.match(r"\s+(\w+)\s+n=(\d+)\s+mean=(\S+)\s+median=(\S+)", line)
    if m and cur is not None:
        cur[m.group(1)] = float(m.group(4))
print(txt.split("\n# one_config")[0].count("[pmc]") - 1, "rocprofv3 --pmc passes, one counter each (tools/pmc_mem_counters.sh); medians per dispatch\n")
for name in ("col_pass_kernel", "row_pass_kernel", "col_pass_staged_kernel", "row_pass_pair_kernel", "row_pass_wave_kernel"):
    c = kern.get(name)
    if not c:
        continue
    g = lambda k: c.get(k, float("nan"))
    cyc = g("GRBM_GUI_ACTIVE") / 8
    print(f"== {name}  (GRBM_GUI_ACTIVE / 8 XCDs = {cyc:,.0f} cycles per dispatch)")
    print(f"  EA reads   {g('TCC_EA0_RDREQ_sum'):12,.0f} requests x 128 B = {g('TCC_EA0_RDREQ_sum') * 128 / 1e6:7.1f} MB   "
          f"(all 'destined for DRAM': {g('TCC_EA0_RDREQ_DRAM_sum'):,.0f}; the counter names the address space, not an Infinity-Cache miss)")
    print(f"  EA writes  {g('TCC_EA0_WRREQ_sum'):12,.0f} requests x  64 B = {g('TCC_EA0_WRREQ_sum') * 64 / 1e6:7.1f} MB")
    print(f"  mean EA read latency   {g('TCC_EA0_RDREQ_LEVEL_sum') / g('TCC_EA0_RDREQ_sum'):7.0f} cycles   (RDREQ_LEVEL / RDREQ)")
    print(f"  mean EA write latency  {g('TCC_EA0_WRREQ_LEVEL_sum') / g('TCC_EA0_WRREQ_sum'):7.0f} cycles   (WRREQ_LEVEL / WRREQ)")
    print(f"  TCP->TCC read latency  {g('TCP_TCC_READ_REQ_LATENCY_sum') / g('TCP_TCC_READ_REQ_sum'):7.0f} cycles,  write (acknowledge) latency {g('TCP_TCC_WRITE_REQ_LATENCY_sum') / g('TCP_TCC_WRITE_REQ_sum'):5.0f} cycles")
    for k, inst, what in (("TCC_EA0_WRREQ_STALL_sum", 128, "L2 channel could not send a write request"),
                          ("TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum", 128, "... because it was out of DRAM write credits"),
                          ("TCC_TOO_MANY_EA_WRREQS_STALL_sum", 128, "... because its own pending-write limit was reached"),
                          ("TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum", 128, "L2 channel out of DRAM read credits"),
                          ("TCC_TAG_STALL_sum", 128, "L2 tag pipeline stalled"),
                          ("TCC_BUSY_sum", 128, "L2 channel has a request pending"),
                          ("TCP_PENDING_STALL_CYCLES_sum", 256, "vector L1 stalled on data pending from L2"),
                          ("TCP_TCP_TA_DATA_STALL_CYCLES_sum", 256, "vector L1 stalled on the TA data interface"),
                          ("TA_TA_BUSY_sum", 256, "address unit busy"),
                          ("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 256, "address path stalled by the cache"),
                          ("TA_DATA_STALLED_BY_TC_CYCLES_sum", 256, "data path stalled by the cache")):
        if k in c:
            print(f"  {k:42s} {c[k]:14,.0f} = {c[k] / inst:9,.0f} per instance = {100 * c[k] / inst / cyc:5.1f} % of the dispatch   ({what})")
    print(f"  L2: {g('TCC_HIT_sum'):,.0f} hits / {g('TCC_MISS_sum'):,.0f} misses of {g('TCC_REQ_sum'):,.0f} requests "
          f"({g('TCC_READ_sum'):,.0f} reads, {g('TCC_WRITE_sum'):,.0f} writes, {g('TCC_STREAMING_REQ_sum'):,.0f} streaming)")
    print()
print("---- raw per-dispatch table ----")
print(txt[txt.index("# one_config"):])
