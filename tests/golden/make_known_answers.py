#!/usr/bin/env python3
"""Re-derive the first-principles entries of known_answers.json (float64, numpy
only) and print them next to the committed values.  The entries that record
reference behaviour (Q1, Q9, filter response) come from SURVEY.md and are not
regenerated here: the reference cannot be built in this image."""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def fir(n):
    i = np.arange(n)
    m = i - n // 2
    out = np.zeros(n)
    odd = (m % 2) != 0
    a = np.pi * m[odd] / n
    out[odd] = -(2.0 / n) / np.tan(a) * 0.5 * (1 - np.cos(2 * np.pi * i[odd] / n))
    return out


if __name__ == "__main__":
    ka = json.load(open(os.path.join(HERE, "known_answers.json")))
    for n, v in ka["fir_abs_sum"].items():
        if n.startswith("_"):
            continue
        print("sum|fir| N=%s: committed %.4f derived %.4f" % (n, v, np.abs(fir(int(n))).sum()))
    f = fir(3072)
    for hz in (20, 50, 100, 1000, 23000):
        w = 2 * np.pi * hz / 48000
        h = np.sum(f * np.exp(-1j * w * (np.arange(3072) - 1536)))
        print("|H|(%5d Hz) = %.4f  phase %+.2f deg" % (hz, abs(h), np.degrees(np.angle(h))))
