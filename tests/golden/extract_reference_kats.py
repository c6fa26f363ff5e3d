#!/usr/bin/env python3
"""Extract the reference's known-answer vectors into tests/golden/reference_kats.json.

Run in the BUILD container only (it reads /root/reference, which does not exist
on the GPU box):

    python tests/golden/extract_reference_kats.py

The reference (Rust) cannot be compiled in this image, so instead of running it
we lift the literal input/expected pairs out of its own `#[cfg(test)]` modules
(src/n_to_bits.rs:408-470, src/n_to_bits2.rs:270-299) and the two bench input
generators (benches/bench_n_to_bits.rs:68-82).  The output is pure data: for
every `assert_eq!(f(args), expected)` one record {fn, input, expected, source}.
"""
import json
import os
import re
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "reference_kats.json")

ASSERT = re.compile(r"assert_eq!\(\s*(\w+)\((.*?)\),\s*(vec!\[.*?\]|\".*?\"\.as_bytes\(\))\s*\);", re.S)


def parse_bytes(tok):
    tok = tok.strip()
    m = re.fullmatch(r'b"([^"]*)"', tok)
    if m:
        return m.group(1)
    m = re.fullmatch(r'"([^"]*)"\.as_bytes\(\)', tok)
    if m:
        return m.group(1)
    raise ValueError(tok)


def parse_words(tok):
    tok = tok.strip()
    m = re.fullmatch(r"&?vec!\[(.*)\]", tok, re.S)
    if not m:
        raise ValueError(tok)
    return [int(x.strip().replace("_", ""), 0) for x in m.group(1).split(",") if x.strip()]


def split_args(s):
    depth, cur, out = 0, "", []
    for ch in s:
        if ch in "([":
            depth += 1
        elif ch in ")]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur)
            cur = ""
        else:
            cur += ch
    out.append(cur)
    return [a.strip() for a in out]


def main():
    records = []
    for rel in ("src/n_to_bits.rs", "src/n_to_bits2.rs"):
        text = open(os.path.join(REF, rel)).read()
        for m in ASSERT.finditer(text):
            fn, args, expected = m.group(1), split_args(m.group(2)), m.group(3)
            line = text.count("\n", 0, m.start()) + 1
            rec = {"fn": fn, "source": "%s:%d" % (rel, line)}
            if fn.startswith("n_to_bits"):
                rec["kind"] = "encode"
                rec["input_ascii"] = parse_bytes(args[0])
                rec["expected_words_hex"] = ["0x%016X" % w for w in parse_words(expected)]
            else:
                rec["kind"] = "decode"
                rec["input_words_hex"] = ["0x%016X" % w for w in parse_words(args[0])]
                rec["len"] = int(args[1])
                rec["expected_ascii"] = parse_bytes(expected)
            records.append(rec)
    # bench generators: b"ATCG".repeat(10000) / b"ATCGN".repeat(8000)
    bench = open(os.path.join(REF, "benches/bench_n_to_bits.rs")).read()
    gens = []
    for name, unit in re.findall(r"fn (get_nucleotides\w*)\(repeat: usize\) -> Vec<u8> \{\s*b\"(\w+)\"\.repeat\(repeat\)", bench):
        gens.append({"generator": name, "unit": unit})
    reps = {"get_nucleotides": 10000, "get_nucleotides_undetermined": 8000}
    for g in gens:
        g["repeat"] = reps[g["generator"]]
        g["source"] = "benches/bench_n_to_bits.rs:68-82"
    doc = {
        "_comment": "Known-answer vectors lifted from the reference's own unit tests by "
        "tests/golden/extract_reference_kats.py; data only.",
        "kats": records,
        "bench_inputs": gens,
    }
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1)
        f.write("\n")
    print("wrote %d KATs, %d bench generators -> %s" % (len(records), len(gens), OUT))
    return 0


if __name__ == "__main__":
    sys.exit(main())
