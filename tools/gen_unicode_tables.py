#!/usr/bin/env python3
"""Generate this repo's Unicode class tables (build container only).

The reference classifies a code point with a two-stage lookup into generated Unicode 11.0.0 data
(``gettyperecord``, reference ``latok/core/src/latok/latok.c:15-29``; data ``latok.h:66-570,1815-4173``) and turns the
flag word into the 12 base feature columns (``latok.c:87-98``).  We do NOT take those arrays: we run every code point
0..0x10FFFF through the reference's own compiled ``_gen_parse_matrix`` (``oracle/_ref``, see oracle/Makefile) and
record the 12 base feature bits it produces.  Only 17 distinct 12-bit words exist, so we re-compress into our own
layouts:

* ``latok_amd/csrc/unicode_tables.inc``  -- device layout: ``kStage1[8704]`` (u8 block id per 128 code points; only 255
  distinct blocks exist so it fits a byte), ``kStage2[255*128]`` (u8 class id 0..16), ``kClassWord[17]`` (12-bit base
  feature word per class, bit i = reference column i of ``latok/core/offsets.py:24-35``) and ``kClassCode[17]`` (the
  8-bit sparse "split code" the fused kernel bit-slices; layout documented in DESIGN.md / split_code.h).
* ``oracle/latok_oracle_tables.inc``     -- oracle layout: sorted run-length list ``{first_cp, word}`` (binary search),
  deliberately a different structure from the device tables so the two cross-check each other.
* ``tests/golden/unicode_classes.json``  -- SHA-256 of the full 0x110000-entry uint16 word array + sample code points
  per class, so the GPU box (no reference there) can pin both layouts.

Code points >= 0x110000 cannot be put in a Python str; the reference maps them to record 0 whose flags are 0
(``latok.c:20-21``, ``latok.h:67``), i.e. word 0 -- both layouts encode that explicitly.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import ref_loader  # noqa: E402

N_CP = 0x110000
SHIFT = 7
NAMES = ["ALPHA", "ALPHA_NUM", "NUM", "LOWER", "UPPER", "SPACE", "SYMBOL", "TWITTER", "CHAR_AT", "CHAR_COLON",
         "CHAR_SLASH", "CHAR_PERIOD"]

# bit positions inside the 12-bit base word (== reference column ids)
A, AN, N, L, U, S, Y, T, AT, CO, SL, PE = range(12)

# 8-bit sparse split code (see latok_amd/csrc/split_code.h).  N is not used by the split rules (only via ALPHA_NUM),
# so classes that differ only in N share a code.
#   bit0 SPACE  bit1 SYMBOL  bit2 LOWER  bit3 UPPER  bit4 ALPHA_NUM  bit5 ALPHA (when SYMBOL=0) / sub0 (when SYMBOL=1)
#   bit6 sub1   bit7 sub2      sub = 0 none, 1 twitter-only(# $ ^), 3 '@' (twitter+at), 4 ':', 5 '/', 6 '.'
def split_code(w: int) -> int:
    b = lambda i: (w >> i) & 1  # noqa: E731
    c = b(S) | (b(Y) << 1) | (b(L) << 2) | (b(U) << 3) | (b(AN) << 4)
    if b(Y):
        assert not b(A) and not b(AN)
        sub = 0
        if b(T) and b(AT):
            sub = 3
        elif b(T):
            sub = 1
        elif b(CO):
            sub = 4
        elif b(SL):
            sub = 5
        elif b(PE):
            sub = 6
        else:
            assert not b(AT)
        c |= sub << 5
    else:
        assert not (b(T) or b(AT) or b(CO) or b(SL) or b(PE))
        c |= b(A) << 5
    return c


# Rule code (runtime rule tables, kModeRules): the split code plus NUM in bit 6 for non-symbols, so that all 12 base
# features of a character can be recovered from one byte (lane_math.h:lk_feature_planes).
def rule_code(w: int) -> int:
    c = split_code(w)
    if not (w >> Y) & 1:
        c |= ((w >> N) & 1) << 6
    return c


def rule_code_word(c: int) -> int:
    """Inverse of rule_code (what the kernel decodes); used to check that the byte loses nothing."""
    b = lambda i: (c >> i) & 1  # noqa: E731
    y = b(1)
    w = (b(5) & (1 - y)) << A | b(4) << AN | (b(6) & (1 - y)) << N | b(2) << L | b(3) << U | b(0) << S | y << Y
    w |= (b(5) & y & (1 - b(7))) << T | (b(5) & b(6) & y) << AT | (b(7) & (1 - b(6)) & (1 - b(5))) << CO
    w |= (b(7) & b(5)) << SL | (b(7) & b(6)) << PE
    return w


def sweep_reference() -> np.ndarray:
    ext = ref_loader.load_ref_ext()
    words = np.zeros(N_CP, np.uint16)
    step = 1 << 16
    weights = (1 << np.arange(12)).astype(np.uint16)
    for c0 in range(0, N_CP, step):
        text = "".join(map(chr, range(c0, c0 + step)))
        m = ext._gen_parse_matrix(text)  # int8[step, 25]; columns 0..11 are context-free
        words[c0:c0 + step] = (m[:, :12].astype(np.uint16) * weights).sum(axis=1)
    return words


def c_array(name, ctype, values, per_line=24):
    out = [f"static const {ctype} {name}[{len(values)}] = {{"]
    for i in range(0, len(values), per_line):
        out.append("    " + ", ".join(str(int(v)) for v in values[i:i + per_line]) + ",")
    out.append("};")
    return "\n".join(out)


def main():
    words = sweep_reference()
    classes = sorted(set(int(w) for w in np.unique(words)))
    assert classes[0] == 0 and len(classes) == 17, classes
    cls_of = {w: i for i, w in enumerate(classes)}
    cls = np.vectorize(cls_of.get, otypes=[np.uint8])(words)

    # ---- device two-stage layout
    nb = N_CP >> SHIFT
    blocks, stage1 = {}, np.zeros(nb, np.uint8)
    stage2 = []
    for b in range(nb):
        key = cls[b << SHIFT:(b + 1) << SHIFT].tobytes()
        if key not in blocks:
            assert len(blocks) < 256
            blocks[key] = len(blocks)
            stage2.append(np.frombuffer(key, np.uint8))
        stage1[b] = blocks[key]
    stage2 = np.concatenate(stage2)
    # out-of-range code points index stage1[nb]: must be an all-class-0 block
    zero_block = blocks.get(bytes(1 << SHIFT))
    assert zero_block is not None
    assert stage1[0] == 0, "ASCII block must be block 0 (fused kernel fast path)"
    codes = [split_code(w) for w in classes]
    rcodes = [rule_code(w) for w in classes]
    for w, c in zip(classes, rcodes):
        assert rule_code_word(c) == w, (w, c)

    dev = [
        "// GENERATED by tools/gen_unicode_tables.py -- do not edit.",
        "// Source of truth: the reference's compiled _gen_parse_matrix swept over 0..0x10FFFF (Unicode 11.0.0 data,",
        "// reference latok/core/src/latok/latok.c:15-29,87-98).  Layout is ours: shift 7, 8-bit stage-1, 8-bit class ids.",
        f"#define LATOK_TBL_SHIFT {SHIFT}",
        f"#define LATOK_TBL_STAGE1_LEN {nb + 1}   /* last entry = block for cp >= 0x110000 */",
        f"#define LATOK_TBL_NBLOCKS {len(blocks)}",
        f"#define LATOK_TBL_NCLASSES {len(classes)}",
        c_array("kStage1", "unsigned char", list(stage1) + [zero_block]),
        c_array("kStage2", "unsigned char", stage2),
        "/* 12-bit base feature word per class: bit i = reference column i (offsets.py:24-35) */",
        c_array("kClassWord", "unsigned short", classes),
        "/* 8-bit sparse split code per class (split_code.h) */",
        c_array("kClassCode", "unsigned char", codes),
        "/* 8-bit rule code per class: split code + NUM in bit 6 for non-symbols (runtime rule tables) */",
        c_array("kClassRuleCode", "unsigned char", rcodes),
        "",
    ]
    with open(os.path.join(ROOT, "latok_amd", "csrc", "unicode_tables.inc"), "w") as f:
        f.write("\n".join(dev))

    # ---- oracle run-length layout
    change = np.nonzero(words[1:] != words[:-1])[0] + 1
    starts = np.concatenate([[0], change])
    orc = [
        "/* GENERATED by tools/gen_unicode_tables.py -- do not edit.  TEST INFRASTRUCTURE (oracle) ONLY.",
        "   Run-length form of the reference's per-code-point 12 base features (latok.c:15-29,87-98): run i covers",
        "   [kRunStart[i], kRunStart[i+1]) with base-feature word kRunWord[i]; code points >= 0x110000 -> word 0. */",
        f"#define ORACLE_NRUNS {len(starts)}",
        c_array("kRunStart", "unsigned int", starts),
        c_array("kRunWord", "unsigned short", words[starts]),
        "",
    ]
    with open(os.path.join(ROOT, "oracle", "latok_oracle_tables.inc"), "w") as f:
        f.write("\n".join(orc))

    # ---- golden pin
    samples = {}
    for w in classes:
        idx = np.nonzero(words == w)[0]
        pick = sorted(set(int(idx[k]) for k in (0, len(idx) // 2, len(idx) - 1)))
        samples[f"{w:#05x}"] = {"features": [NAMES[i] for i in range(12) if w >> i & 1], "count": int(len(idx)),
                                "code_points": pick}
    golden = {
        "generator": "tools/gen_unicode_tables.py",
        "source": "reference _gen_parse_matrix (oracle/_ref) swept over every code point 0..0x10FFFF",
        "n_code_points": N_CP,
        "sha256_uint16le_words": hashlib.sha256(words.astype("<u2").tobytes()).hexdigest(),
        "n_classes": len(classes),
        "n_runs": int(len(starts)),
        "classes": samples,
    }
    with open(os.path.join(ROOT, "tests", "golden", "unicode_classes.json"), "w") as f:
        json.dump(golden, f, indent=1)
    print(f"classes={len(classes)} blocks={len(blocks)} runs={len(starts)} sha={golden['sha256_uint16le_words'][:16]}…")


if __name__ == "__main__":
    main()
