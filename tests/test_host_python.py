"""Host-side Python mirror (no GPU): the canonical proof encoding and the transcript layout."""
import json
import os

from util import R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_proof_bytes_round_trip():
    """Proof.from_bytes lays the record out as include/sonic_hip.h says (Proof then HscProof field order,
    src/Sonic/Protocol.hs:28-38, src/Sonic/Signature.hs:22-29); cross-checked with the oracle's decoder"""
    import sonic_amd
    from oracle import pairing as pg
    for c in json.load(open(os.path.join(GOLD, "prove_small.json")))["cases"]:
        raw = bytes.fromhex(c["proof"])
        Q = c["Q"]
        assert len(raw) == (7 + 4 * Q) * 96 + (5 + 2 * Q) * 32
        p = sonic_amd.Proof.from_bytes(raw, Q)
        o = pg.proof_from_bytes(raw, Q)
        assert (p.prR, p.prT, p.prA, p.prWa, p.prB, p.prWb, p.prWt, p.prS) == tuple(o[k] for k in ("prR", "prT", "prA", "prWa", "prB", "prWb", "prWt", "prS"))
        h, oh = p.prHscProof, o["prHscProof"]
        assert h.hscS == oh["hscS"] and h.hscW == oh["hscW"] and (h.hscQv, h.hscC, h.hscU, h.hscV) == (oh["hscQv"], oh["hscC"], oh["hscU"], oh["hscV"])
        tr = [int(v, 16) for v in c["transcript"]]
        assert h.hscU == tr[6 + 2 * Q] and h.hscV == tr[7 + 2 * Q]      # draw order: cns, y, z, ys, zs, u, v
        assert p.to_bytes() == raw


def test_hsc_proof_bytes_round_trip():
    """the HscProof part of a proof is what sonic_prover_hsc_prove writes and sonic_hsc_verify reads (Signature.hs:22-29)"""
    from sonic_amd.protocol import _hsc_from_bytes, _hsc_to_bytes
    import sonic_amd
    c = json.load(open(os.path.join(GOLD, "prove_small.json")))["cases"][0]
    raw, Q = bytes.fromhex(c["proof"]), c["Q"]
    tail = raw[576:]
    h = _hsc_from_bytes(tail, Q)
    assert _hsc_to_bytes(h) == tail and h == sonic_amd.Proof.from_bytes(raw, Q).prHscProof
    assert len(tail) == (2 + 4 * Q) * 96 + (2 + 2 * Q) * 32
    import pytest
    with pytest.raises(ValueError):
        _hsc_from_bytes(tail[:-1], Q)


def test_transcript_draws():
    import random
    from sonic_amd.protocol import draw_transcript, transcript_len
    assert transcript_len(3) == 14
    t = draw_transcript(2, random.Random(1))
    assert len(t) == 12 and all(0 <= v < R for v in t)
    assert len(draw_transcript(1)) == 10
