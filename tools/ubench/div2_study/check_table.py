#!/usr/bin/env python3
"""CPU cross-check of tools/ubench/div2_study/div2_table.bin (written by div2_exhaustive.hip on an MI355X): for sampled divisor
significands of every named variant, the two-operation quotient q = fma(n, zh, RN(n zl)) equals IEEE division for ALL 2^23
numerator significands (host fmaf); a divisor outside the table is exact with the plain pair; a pair-less one fails on exactly
one numerator.  This check is what caught the first version of the enumeration, which had used the candidate zh (not RN(1/d))
inside its 3-operation reference and vouched for pairs that are not exact.   python tools/ubench/div2_study/check_table.py"""
import os
import subprocess
import tempfile

import numpy as np


def main():

    """nanocall_amd/data/div2_table.bin (embedded in the library; viterbi_kernel.hip): sorted, one entry per failing divisor
    significand, variants 1..24 or 255 -- and a C check of the claim itself on a sample: for divisors with a named variant the
    two-operation quotient equals IEEE division for ALL 2^23 numerator significands (tools/ubench/div2_exhaustive.hip did
    every divisor on the GPU; this re-does a few on the CPU with the host's fmaf)."""
    t = np.fromfile(os.path.join(os.path.dirname(os.path.abspath(__file__)), "div2_table.bin"), dtype="<u4")
    sig, code = t >> 8, t & 255
    assert len(t) == 106762 and (np.diff(sig.astype(np.int64)) > 0).all() and sig.max() < (1 << 23)
    assert set(np.unique(code)) <= set(range(1, 25)) | {255} and int((code == 255).sum()) == 5323
    src = r'''
#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
static float ulps(float v, int k) { uint32_t b; memcpy(&b, &v, 4); b += (uint32_t)k; memcpy(&v, &b, 4); return v; }
int main(int argc, char** argv) {
    long bad = 0;
    for (int a = 1; a + 1 < argc; a += 2) {
        uint32_t m = strtoul(argv[a], 0, 10); unsigned v = strtoul(argv[a + 1], 0, 10);
        uint32_t bits = (127u << 23) | m; float d; memcpy(&d, &bits, 4);
        const int order[5] = {0, -1, 1, -2, 2};
        int dzh = order[v / 5], dzl = order[v % 5];
        float zh = ulps(1.0f / d, dzh);
        float zl = fmaf(-zh, d, 1.0f) / d;
        if (dzl) zl = zl >= 0.0f ? ulps(zl, dzl) : ulps(zl, -dzl);
        for (uint32_t nm = 0; nm < (1u << 23); ++nm) {
            uint32_t nb = (127u << 23) | nm; float n; memcpy(&n, &nb, 4);
            float q = fmaf(n, zh, n * zl);
            if (q != n / d) ++bad;
        }
    }
    printf("%ld\n", bad);
    return 0;
}
'''
    rng = np.random.default_rng(5)
    picks = []
    for v in sorted(set(np.unique(code).tolist()) - {255}):      # every named variant, two divisors each: exact for every numerator
        pool = sig[code == v]
        picks += [(int(m), v) for m in rng.choice(pool, size=min(2, len(pool)), replace=False)]
    plain = [m for m in rng.integers(0, 1 << 23, 40).tolist() if m not in set(sig.tolist())][:3]
    picks += [(int(m), 0) for m in plain]                   # not in the table: the plain pair is exact
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "t.c"), "w").write(src)
        subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-mfma", "t.c", "-o", "t", "-lm"], cwd=d, check=True)
        out = subprocess.run([os.path.join(d, "t")] + [str(x) for p in picks for x in p], capture_output=True, text=True, check=True)
        assert int(out.stdout.strip()) == 0, out.stdout
        # and the other way round: a pair-less divisor does fail with its plain pair (exactly one numerator)
        m = int(sig[code == 255][0])
        out = subprocess.run([os.path.join(d, "t"), str(m), "0"], capture_output=True, text=True, check=True)
        assert int(out.stdout.strip()) == 1
    print("div2_table.bin: %d entries, %d pair-less; sampled variants exact on all 2^23 numerators" % (len(t), int((code == 255).sum())))


if __name__ == "__main__":
    main()
