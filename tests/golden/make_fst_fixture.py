#!/usr/bin/env python3
"""A small decoding graph for the refmodel fixture, in the reference's FST file format.

    python3 tests/golden/make_fst_fixture.py        ->  tests/golden/refmodel/wordloop.fst

OUR synthetic graph (nothing of the reference's data), written in the layout Fst::Read parses
(fst.cc:30-90): a 32-byte section name "pk::fst_0", i32 section size, i32 states, i32 arcs,
i32 start state, float final[states], i32 first-arc index[states], then the arcs as
{i32 next_state, i32 input_label (transition-id, 0 = epsilon), i32 output_label (word id,
0 = none), float weight} (fst.h:16-21).  tests/test_oracle_fixtures.py parses the reference's own
test/data/testinput.fst facts (test/fst_test.cc:23-62) with the same reader to pin the layout.

Graph: a loop of NUM_WORDS words, three emitting HMM-like states per word (self-loop + forward
arc each, transition-ids 1..36 of the refmodel's 57), an epsilon arc back to the loop state.
"""
import os
import struct

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
NUM_WORDS = 6
INF = float("inf")


def build():
    """-> (final[], arcs_by_state[]) with arcs as (next, ilabel, olabel, weight)."""
    rng = np.random.default_rng(0xF57)
    n_states = 1 + 3 * NUM_WORDS
    final = [INF] * n_states
    final[0] = 0.5
    arcs = [[] for _ in range(n_states)]
    tid = 1
    for w in range(1, NUM_WORDS + 1):
        a = 1 + 3 * (w - 1)
        fwd, loop = [tid, tid + 1, tid + 2], [tid + 3, tid + 4, tid + 5]
        tid += 6
        arcs[0].append((a, fwd[0], w, float(np.float32(np.log(NUM_WORDS) + 0.05 * w))))
        for k in range(3):
            s = a + k
            arcs[s].append((s, loop[k], 0, float(np.float32(rng.uniform(0.2, 0.6)))))
            if k < 2:
                arcs[s].append((s + 1, fwd[k + 1], 0, float(np.float32(rng.uniform(0.6, 1.2)))))
        arcs[a + 2].append((0, 0, 0, float(np.float32(0.1))))       # epsilon, back to the loop state
        final[a + 2] = 2.0
    return final, arcs


def write(path):
    final, arcs = build()
    flat, first = [], []
    for out in arcs:
        first.append(len(flat) if out else -1)
        flat.extend(out)
    body = struct.pack("<iii", len(final), len(flat), 0)
    body += struct.pack("<%df" % len(final), *final)
    body += struct.pack("<%di" % len(first), *first)
    for nxt, il, ol, wt in flat:
        body += struct.pack("<iiif", nxt, il, ol, wt)
    with open(path, "wb") as f:
        f.write(b"pk::fst_0".ljust(32, b"\0"))
        f.write(struct.pack("<i", len(body)))
        f.write(body)
    return len(final), len(flat)


if __name__ == "__main__":
    out = os.path.join(HERE, "refmodel", "wordloop.fst")
    print(out, "states/arcs:", write(out))
