"""records_in_image_order: the record stream of the one-launch decode (records in arrival order)
moved into ganet_decode_pack's image order.  Host code only."""
import numpy as np

from ganet_amd.decode import records_in_image_order


def _stream(words, order):
    B = len(words)
    off, pos = np.zeros(B, np.int64), 0
    for b in order:
        off[b] = pos
        pos += words[b]
    rec = np.zeros(pos, np.int32)
    for b in range(B):
        rec[off[b]:off[b] + words[b]] = 1000 * b + np.arange(words[b])
    hdr = np.zeros((B, 8), np.int32)
    hdr[:, 0] = np.arange(B) + 1
    hdr[:, 4], hdr[:, 5] = off, words
    hdr[0, 6] = pos
    return hdr, rec


def test_any_arrival_order_gives_the_image_order_stream():
    rng = np.random.default_rng(0)
    for _ in range(20):
        B = int(rng.integers(1, 40))
        words = rng.integers(0, 30, B)
        want_h, want_r = _stream(words, range(B))
        h, r = _stream(words, rng.permutation(B))
        h0 = h.copy()
        h2, r2 = records_in_image_order(h, r)
        assert np.array_equal(h2, want_h) and np.array_equal(r2, want_r)
        assert np.array_equal(h, h0)            # the caller's header is not rewritten
    h, r = _stream([3, 4], [0, 1])
    h2, r2 = records_in_image_order(h, r)
    assert h2 is h and r2 is r                   # already in order: nothing is copied


def test_a_stream_with_a_record_that_did_not_fit_is_left_alone():
    h, r = _stream([3, 4, 2], [2, 0, 1])
    h[1, 3] |= 8
    h2, r2 = records_in_image_order(h, r)
    assert h2 is h and r2 is r
