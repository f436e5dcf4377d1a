#!/usr/bin/env python3
"""Which stage-kernel instantiations does a run actually launch?  ("record which kernels the 51 e2e
cases ... actually launch (the rocprofv3 name set)".)

    rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o kt -- python -m pytest tests/... -m gpu -q
    python tools/kernel_names.py --summarise <dir> [<dir> ...] --md profiles/r05_kernels_launched.md [--lib dpm_solver_amd/libdpm_hip.so]

Reads every *kernel_stats.csv under the directories, keeps the kernels of this library (stage_kernel, stage_kernel_multi,
stage_kernel_scalar, stage_thresh_kernel and the double-precision ones), and writes: the distinct instantiations launched
against the number the library holds (nm), and per family the template arguments that were reached, with call counts.
"""
import argparse
import collections
import csv
import glob
import os
import re
import subprocess

FAMILIES = ("stage_kernel_multi", "stage_kernel_scalar", "stage_kernel_f64", "stage_thresh_kernel_f64", "stage_thresh_kernel",
            "stage_kernel")
FORMS = {"0": "LIN1", "1": "TWO", "2": "MS3", "3": "SS3T", "4": "DENOISE", "-1": "run-time"}
GUIDES = {"0": "-", "1": "cfg", "2": "classifier", "-1": "run-time"}
SPECS = {"1": "noise ++ (compile-time)", "100": "run-time prologue", "0": "noise eps (compile-time)"}


def family_of(name):
    for f in FAMILIES:
        if re.search(r"\b%s\b" % f, name):
            return f
    return None


def base(name):
    """`family<template arguments>` of a demangled kernel name (the argument list and any prefix dropped)"""
    fam = family_of(name)
    i = re.search(r"\b%s\b" % fam, name).end()
    if i >= len(name) or name[i] != "<":
        return fam
    depth = 0
    for j in range(i, len(name)):
        depth += name[j] == "<"
        depth -= name[j] == ">"
        if depth == 0:
            return fam + name[i:j + 1]
    return fam + name[i:]


def targs(name):
    m = re.search(r"<(.*)>", name)
    return [a.strip() for a in m.group(1).split(",")] if m else []


def describe(fam, a):
    a = [x.replace("(anonymous namespace)::", "").replace("dpmk::", "") for x in a]
    if fam == "stage_kernel" and len(a) >= 9:
        d = "%s/%s %s guide %s%s, %s%s%s" % (a[0], a[1], FORMS.get(a[2], a[2]), GUIDES.get(a[3], a[3]), " xe" if a[4] == "true" else "",
                                            SPECS.get(a[5], a[5]), ", KExt" if a[8] == "true" else "",
                                            ", device-resident coefficients" if len(a) > 9 and a[9] == "true" else "")
        return d + " (U %s, nt %s)" % (a[6], a[7])
    if fam == "stage_kernel_multi" and len(a) >= 5:
        return "%s/%s %s guide %s, %s" % (a[0], a[1], FORMS.get(a[2], a[2]), GUIDES.get(a[3], a[3]), SPECS.get(a[4], a[4]))
    if fam == "stage_thresh_kernel" and len(a) >= 7:
        return "%s/%s %s guide %s%s, HOT %s" % (a[0], a[1], FORMS.get(a[2], a[2]), GUIDES.get(a[3], a[3]),
                                                " xe" if a[4] == "true" else "", a[6])
    return ", ".join(a)


def library_count(lib):
    out = subprocess.run(["nm", "-C", "--defined-only", lib], stdout=subprocess.PIPE, text=True, check=True).stdout
    names = set()
    for l in out.splitlines():
        m = re.match(r"^[0-9a-f]+ d (.*)$", l)
        if m and family_of(m.group(1)):
            names.add(base(m.group(1)))
    return len(names)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--summarise", nargs="+", required=True)
    ap.add_argument("--md", required=True)
    ap.add_argument("--lib", default=os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "dpm_solver_amd", "libdpm_hip.so"))
    ap.add_argument("--title", default="stage kernels launched")
    a = ap.parse_args()
    calls = collections.Counter()
    for d in a.summarise:
        for f in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                name = row.get("Name") or row.get("KernelName") or ""
                if family_of(name):
                    calls[base(name)] += int(float(row.get("Calls", 0) or 0))
    fam = collections.defaultdict(list)
    for n, c in calls.items():
        fam[family_of(n)].append((describe(family_of(n), targs(n)), c))
    total = library_count(a.lib) if os.path.exists(a.lib) else None
    with open(a.md, "w") as f:
        f.write("# %s\n\n" % a.title)
        f.write("**%d distinct stage-kernel instantiations launched**%s, %d launches in all.\n\n" % (
            len(calls), " of the %d the library holds" % total if total else "", sum(calls.values())))
        for k in FAMILIES:
            if k not in fam:
                continue
            f.write("## %s: %d instantiations, %d launches\n\n| instantiation | launches |\n|---|---|\n" % (k, len(fam[k]), sum(c for _, c in fam[k])))
            for d_, c in sorted(fam[k], key=lambda t: (-t[1], t[0])):
                f.write("| %s | %d |\n" % (d_, c))
            f.write("\n")
    print("%d distinct instantiations, %d launches -> %s" % (len(calls), sum(calls.values()), a.md))


if __name__ == "__main__":
    main()
