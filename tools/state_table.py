#!/usr/bin/env python3
"""The "current state" table of README.md / DESIGN.md section 9 from a bench.py record -- so the documents quote a run, not a memory.

usage: python tools/state_table.py BENCH_rNN.json            (a driver record: its "parsed" object is the bench line)
       python tools/state_table.py gpurun_out/bench_x.json   (bench.py's own JSON line)
Prints a markdown table; --update rewrites the block between the STATE-TABLE markers of README.md and DESIGN.md in place."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(path):
    d = json.load(open(path))
    return d.get("parsed", d), ("driver record %s (head %s)" % (os.path.basename(path), d.get("head", "?")) if "parsed" in d else "bench.py run %s" % os.path.basename(path))


def g(d, *ks, default=None):
    for k in ks:
        if not isinstance(d, dict) or k not in d:
            return default
        d = d[k]
    return d


def fmt(v, nd=2):
    return "-" if v is None else ("%.*f" % (nd, v))


def sci(v):
    if v is None:
        return "-"
    e = len(str(int(v))) - 1
    return "%.2f·10^%d" % (v / 10 ** e, e)


def table(b, src):
    ph = g(b, "roofline", "phase_ms", default={})
    rows = [
        ("**headline: one proof, host bytes to host bytes** (`rv_prove`, 10^7 gates / 5.01·10^6 AND)", "%s AND/s" % sci(b.get("value")), "%s ms" % fmt(b.get("ms_per_step"))),
        ("  GPU phases: masks / interpreter / digests / openings", "", " / ".join(fmt(ph.get(k)) for k in ("masks", "interp", "hash", "open")) + " ms"),
        ("  masks || levels phase vs the integer-VALU issue peak (%s + level launches)" % g(b, "roofline", "kernel", default="?"), "%s of 1.23 T wavefront instructions/s (mix ceiling %s)" % (fmt(g(b, "roofline", "frac")), fmt(g(b, "roofline", "ceiling_frac"))), "%s ms" % fmt(g(b, "roofline", "phase_ms", "interp"))),
        ("  interpreter kernel vs HBM roofline (%s)" % g(b, "roofline", "hbm", "kernel", default="?"), "%s of 8 TB/s" % fmt(g(b, "roofline", "hbm", "frac")), "%s us per launch x %s" % (fmt(g(b, "roofline", "avg_launch_us"), 1), g(b, "roofline", "launches_per_proof"))),
        ("  the level kernel with nothing beside it (`RV_OVERLAP=0 RV_EARLY=0`, same run)", "%s of 8 TB/s" % fmt(g(b, "roofline", "kernel_alone", "frac")), "%s us per launch; such a proof: %s ms" % (fmt(g(b, "roofline", "kernel_alone", "avg_launch_us"), 1), fmt(g(b, "roofline", "kernel_alone", "ms_per_proof_host_to_host")))),
        ("  level launches + mask generator beside them, both kernels' bytes over the phase", "%s of 8 TB/s" % fmt(g(b, "roofline", "concurrent", "phase_frac_incl_masks")), ""),
        ("  issue fraction, mask generator + level launches / transcript hashes", "", "%s / %s" % (fmt(g(b, "roofline", "valu", "kernels", "masks+interp", "issue_frac", default=g(b, "roofline", "valu", "kernels", "masks", "issue_frac"))), fmt(g(b, "roofline", "valu", "kernels", "hash", "issue_frac")))),
        ("all-AND variant (10^7 AND)", "%s AND/s" % sci(g(b, "all_and", "value")), "%s ms" % fmt(g(b, "all_and", "ms_per_proof"))),
        ("openings left in HBM (`rv_prove_device`)", "%s AND/s" % sci(g(b, "device_resident", "value")), "%s ms" % fmt(g(b, "device_resident", "ms_per_proof"))),
        ("batch of 8 proofs per call (`rv_prove_batch`)", "%s AND/s" % sci(g(b, "prove_batch_host", "value")), "%s ms per proof" % fmt(g(b, "prove_batch_host", "ms_per_proof"))),
        ("verifier (`rv_verify`, strict)", "%s AND/s" % sci(g(b, "verifier", "value")), "%s ms" % fmt(g(b, "verifier", "ms"))),
        ("first proof of an unseen circuit (`rv_prove_ops`)", "%s AND/s" % sci(g(b, "first_proof", "and_per_s")), "%s ms" % fmt(g(b, "first_proof", "first_proof_ms"), 1)),
        ("  the same call again on the same op list (the context keeps compiled circuits by content)", "%s AND/s" % sci(g(b, "first_proof", "repeat_and_per_s")), "%s ms" % fmt(g(b, "first_proof", "repeat_proof_ms"))),
        ("streaming prover / verifier (`rv_prove_streaming`, `rv_verify_streaming`)", "%s AND/s" % sci(g(b, "streaming", "value")), "%s / %s ms" % (fmt(g(b, "streaming", "ms"), 1), fmt(g(b, "streaming", "verify_streaming", "ms"), 1))),
        ("  the same with NO transcripts kept between the passes (`RV_STREAM_KEEP_MB=0`: wire store + one chunk + the proof on the device)", "%s AND/s" % sci(g(b, "streaming", "bounded_memory", "value")), "%s ms" % fmt(g(b, "streaming", "bounded_memory", "ms"), 1)),
        ("Z64, 10^6 MUL (config 5)", "%s MUL/s" % sci(g(b, "secondary", "z64", "mul_per_s")), "%s ms (verify %s)" % (fmt(g(b, "secondary", "z64", "ms_per_proof")), fmt(g(b, "secondary", "z64", "verify_ms")))),
        ("AES-128 / SHA-256, one proof", "%s / %s AND/s" % (sci(g(b, "secondary", "aes128", "single_proof_and_per_s")), sci(g(b, "secondary", "sha256", "single_proof_and_per_s"))),
         "%s / %s ms" % (fmt(g(b, "secondary", "aes128", "single_proof_ms"), 3), fmt(g(b, "secondary", "sha256", "single_proof_ms"), 3))),
        ("AES-128 / SHA-256, 256 proofs per call", "%s / %s AND/s" % (sci(g(b, "secondary", "aes128", "batch256", "and_per_s")), sci(g(b, "secondary", "sha256", "batch256", "and_per_s"))), ""),
        ("one rank's share of a sharded proof, 256 / 128 / 64 / 32 repetitions", "", " / ".join(fmt(g(b, "shard_ms", "ms", k)) for k in ("256", "128", "64", "32")) + " ms"),
        ("CPU port on the same host (%s threads, `cpu_baseline.kind` = %s)" % (g(b, "cpu_baseline", "cores"), g(b, "cpu_baseline", "kind")), "%s AND/s" % sci(g(b, "cpu_baseline", "value")), ""),
    ]
    par = b.get("parity", {})
    out = ["<!-- STATE-TABLE: generated by tools/state_table.py from %s; do not edit by hand -->" % src, "",
           "| what | rate | time |", "|---|---|---|"]
    out += ["| %s | %s | %s |" % r for r in rows]
    out += ["", "Parity flags of that run: timed proof bit-exact vs the CPU oracle %s, verifies strictly %s, deterministic %s; all-AND bit-exact %s; Z64 full size bit-exact %s." % (
        par.get("timed_proof_bit_exact_vs_cpu"), par.get("last_timed_proof_verifies_strict"), par.get("rv_prove_is_deterministic"),
        g(b, "all_and", "bit_exact_vs_cpu"), g(b, "secondary", "z64", "full_size_bit_exact_vs_cpu")), "", "<!-- /STATE-TABLE -->"]
    return "\n".join(out)


if __name__ == "__main__":
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    b, src = load(args[0])
    t = table(b, src)
    if "--update" in sys.argv:
        for name in ("README.md", "DESIGN.md"):
            p = os.path.join(ROOT, name)
            s = open(p).read()
            s2, n = re.subn(r"<!-- STATE-TABLE:.*?<!-- /STATE-TABLE -->", lambda m: t, s, flags=re.S)
            if n:
                open(p, "w").write(s2)
            print("%s: %d block(s) rewritten" % (name, n))
    else:
        print(t)
