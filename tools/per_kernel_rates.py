#!/usr/bin/env python3
"""Per-kernel table from the three PMC passes of tools/profile_round.sh (profiles/<tag>_pmc.txt): memory-side bytes per
launch (2 x FETCH_SIZE + WRITE_SIZE KiB: the gfx950 correction of the microarchitecture guide), the rate they arrive at,
MFMA pipe busy (SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs) and, for launches of 100 us and
more, the clock the counter pass implies (an upper estimate: the counter also runs during the launch overhead).
Usage: python tools/per_kernel_rates.py profiles/r03_pmc.txt > profiles/r03_per_kernel_rates.txt"""
import re
import sys


def short(name):
    m = re.search(r"dark::(\w+)(<[^>(]*>)?", name)
    if m:
        return m.group(1) + (m.group(2) or "")
    m = re.search(r"_ZN4dark\d+(\w+?)I(\w+?)E", name)
    if m:
        return m.group(1)[:24] + "<" + m.group(2)[:18] + ">"
    return name[:44]


def main():
    txt = open(sys.argv[1]).read()
    tag = re.match(r"# (\w+):", txt).group(1)
    passes = re.split(r"^# ---- pass ", txt, flags=re.M)[1:]
    data = {}
    for sec in passes:
        lines = sec.split("\n")
        i = 1
        while i < len(lines):
            m = re.match(r"\s+dispatches (\d+) of (\d+)\s+avg duration ([\d.]+) us", lines[i]) if i < len(lines) else None
            if m and i >= 1:
                name = short(lines[i - 1])
                d = data.setdefault(name, {})
                d["n"], d["us"] = int(m.group(1)), float(m.group(3))
                j = i + 1
                while j < len(lines) and re.match(r"\s+[A-Z_]+\s+\d+", lines[j]):
                    k, v = lines[j].split()
                    d[k] = float(v)
                    j += 1
                i = j
            else:
                i += 1
    print(f"# {tag}: per kernel (full-size launches of `bench.py --steps 1 --warmup 0`), from the three PMC passes of profiles/{tag}_pmc.txt:")
    print("# memory-side bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE KiB (the gfx950 correction of the microarchitecture guide),")
    print("# rate = bytes / average duration, MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs over GRBM_GUI_ACTIVE / 8 XCDs,")
    print("# clock = GRBM_GUI_ACTIVE / 8 / duration in the counter pass: an UPPER estimate, launches of 100 us and more only")
    print(f"{'kernel':46s} {'launches':>8s} {'avg us':>8s} {'MB/launch':>10s} {'TB/s':>6s} {'MFMA busy':>9s} {'GHz':>5s}")
    rows = sorted((d for d in data.items() if "FETCH_SIZE" in d[1] and "WRITE_SIZE" in d[1]),
                  key=lambda kv: -kv[1]["n"] * kv[1]["us"])
    for name, d in rows[:22]:
        mb = (2 * d["FETCH_SIZE"] + d["WRITE_SIZE"]) * 1024 / 1e6
        tbs = mb / d["us"] if d["us"] else 0.0
        busy = clk = None
        if d.get("GRBM_GUI_ACTIVE"):
            cyc = d["GRBM_GUI_ACTIVE"] / 8
            if d.get("SQ_VALU_MFMA_BUSY_CYCLES"):
                busy = d["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024 / cyc * 100
            if d["us"] >= 100:
                clk = cyc / d["us"] / 1e3
        print(f"{name:46s} {d['n']:8d} {d['us']:8.1f} {mb:10.1f} {tbs:6.2f} "
              f"{(f'{busy:7.1f} %' if busy and busy > 0.5 else '-'):>9s} {(f'{clk:5.2f}' if clk else '-'):>5s}")


if __name__ == "__main__":
    main()
