"""Host side of the index-compressed chunked ELL layout (compress_indices; SellChunk in
ogl_amd/csrc/common.hpp): ogl_host_sell_check builds the layout from a CSR pattern, decodes it the way
k_spmv_sell does and compares with the input.  No GPU needed."""
import numpy as np
import pytest

from ogl_amd import capi, synthetic
from helpers import oracle_csr

CHUNK = 512


def poisson_pattern(oracle, **kw):
    case = synthetic.poisson_block(**kw)
    rp, cols, _ = oracle_csr(oracle, case)
    return rp, cols


@pytest.mark.parametrize("kw", [dict(gx=5, gy=4, gz=3), dict(gx=33, gy=31, gz=29), dict(gx=1, gy=1, gz=1),
                                dict(gx=1031, gy=1, gz=1), dict(gx=64, gy=8, gz=1),
                                dict(gx=6, gy=5, gz=4, periodic_x=True)])
@pytest.mark.parametrize("sym", [True, False])
def test_structured_patterns_qualify_and_decode(oracle, kw, sym):
    rp, cols = poisson_pattern(oracle, symmetric=sym, **kw)
    ok, slots, dict_entries, code_bytes = capi.host_sell_check(rp, cols)
    n, nnz = len(rp) - 1, int(rp[-1])
    n_chunks = (n + CHUNK - 1) // CHUNK
    assert ok
    assert slots % CHUNK == 0 and slots >= nnz
    # every chunk is padded to its own longest row only
    widths = [int(np.diff(rp[c * CHUNK:min(n, (c + 1) * CHUNK) + 1]).max()) for c in range(n_chunks)]
    assert slots == CHUNK * sum(widths)
    # structured patterns take the pattern mode: one byte per row (2 per thread)
    assert code_bytes == 512 * n_chunks
    assert dict_entries <= 2048 * n_chunks


def test_seven_point_box_row_patterns(oracle):
    # 16^3 box: a chunk = two xy-planes; a row's pattern is fixed by which of its 6 neighbours exist
    rp, cols = poisson_pattern(oracle, gx=16, gy=16, gz=16, symmetric=True)
    ok, slots, dict_entries, code_bytes = capi.host_sell_check(rp, cols)
    assert ok and slots == 7 * 16 ** 3 and code_bytes == 512 * (16 ** 3 // CHUNK)
    # interior chunks: 3 x 3 (x, y position classes) patterns of 7 offsets; first and last chunk also
    assert dict_entries % 7 == 0 and 9 * 7 * 8 <= dict_entries <= 27 * 7 * 8


def test_many_row_patterns_fall_back_to_offset_codes():
    # 40 diagonals of which every row uses a pseudo-random subset: > 256 row patterns per chunk, but
    # only 40 distinct offsets -> one byte per (row, slot)
    rng = np.random.default_rng(9)
    n = 1024
    offs = np.arange(-20, 20)
    rows = []
    for r in range(n):
        pick = offs[rng.random(40) < 0.92]
        c = np.unique(np.concatenate([[r], [r + o for o in pick if 0 <= r + o < n]]))
        rows.append(c)
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    ok, slots, dict_entries, code_bytes = capi.host_sell_check(rp, cols)
    widths = [max(len(c) for c in rows[i * CHUNK:(i + 1) * CHUNK]) for i in range(2)]
    assert ok and slots == CHUNK * sum(widths)
    # per chunk: 512 row-length bytes in front of the codes, then 2 x width code bytes per thread
    assert code_bytes == sum(512 + 256 * ((2 * w + 15) // 16 * 16) for w in widths)
    assert dict_entries <= 2 * 41


def random_rows(n, per_row, reach, seed):
    rng = np.random.default_rng(seed)
    rows = []
    for r in range(n):
        lo, hi = max(0, r - reach), min(n, r + reach + 1)
        rows.append(np.unique(np.concatenate([[r], rng.integers(lo, hi, per_row)])))
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    return rp, np.concatenate(rows).astype(np.int32), rows


def test_unstructured_pattern_takes_16_bit_deltas():
    # 6 random neighbours per row: far more than 255 distinct offsets in a chunk, but every distance
    # between consecutive columns of a row fits 16 bits -> 10 bytes per entry instead of CSR's 12
    n = 4096
    rp, cols, rows = random_rows(n, 6, n, 5)
    ok, slots, dict_entries, code_bytes = capi.host_sell_check(rp, cols)
    assert ok and dict_entries == 0
    widths = [max(len(c) for c in rows[i * CHUNK:(i + 1) * CHUNK]) for i in range(n // CHUNK)]
    assert slots == CHUNK * sum(widths)
    # 512 row-length bytes per chunk, then words of 4 slots x 2 rows
    assert code_bytes == sum(512 + 256 * 16 * ((w + 3) // 4) for w in widths)
    assert capi.host_sell_modes(rp, cols) == (True, n // CHUNK, 0)


def test_far_apart_columns_take_32_bit_columns_per_chunk():
    # a band of +-300 around the diagonal plus, in the second chunk only, couplings 70,000 rows away:
    # that chunk cannot use 16-bit deltas and stores plain columns, the others are unaffected
    n = 80_000
    rp, cols, rows = random_rows(n, 4, 300, 6)
    ok, d16_before, c32 = capi.host_sell_modes(rp, cols)
    assert ok and c32 == 0 and d16_before >= n // CHUNK - 1      # (the short last chunk may take 1-byte codes)
    for r in range(CHUNK, 2 * CHUNK, 3):
        rows[r] = np.unique(np.concatenate([rows[r], [r + 70_000]]))
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    ok, d16, c32 = capi.host_sell_modes(rp, cols)
    assert ok and c32 == 1 and d16 == d16_before - 1
    # the spread of the FIRST offsets of a chunk must fit too: rows starting 70,000 columns back
    rows[140 * CHUNK + 1] = np.unique(np.concatenate([rows[140 * CHUNK + 1], [140 * CHUNK + 1 - 3000]]))
    rows[140 * CHUNK + 2] = np.unique(np.concatenate([[2], [8_000], rows[140 * CHUNK + 2]]))
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    assert capi.host_sell_modes(rp, cols) == (True, d16 - 1, 2)


def test_unsorted_rows_take_32_bit_columns():
    # deltas must be >= 0: a row stored out of column order (never produced by the LDU conversion, but
    # the layout is also used for other CSR matrices) falls back to plain columns and still decodes
    n = 1024
    rp, cols, rows = random_rows(n, 5, n, 7)
    k = rp[100]
    cols[k], cols[k + 1] = cols[k + 1], cols[k]
    assert capi.host_sell_modes(rp, cols) == (True, 1, 1)


def test_one_long_row_per_chunk_is_spilled():
    # tridiagonal matrix + a 100-wide row in every chunk: padding every row to 100 slots is out of the
    # question; the chunk keeps 3 slots per row and the long row's tail (98 entries) goes to the spill list
    n = 8192
    rows = []
    for r in range(n):
        c = {max(r - 1, 0), r, min(r + 1, n - 1)}
        if r % CHUNK == 7:
            c |= set(range(r, min(n, r + 100)))
        rows.append(np.array(sorted(c)))
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    ok, read, spilled = capi.host_sell_spilled(rp, cols)
    assert ok and read == 3 * n and spilled == (n // CHUNK) * 98      # 101 entries, 3 stay in the planes
    assert capi.host_sell_check(rp, cols)[1] == 3 * n          # allocated = 3 planes per chunk


def test_a_few_long_rows_among_many_are_spilled_not_padded():
    # hex-dominant mesh stand-in: 7 entries per row, 3 % of the rows have 12: without the spill every
    # wavefront would run to 12 slots (1.7 x the entries)
    rng = np.random.default_rng(12)
    n = 16 * CHUNK
    rows = []
    for r in range(n):
        k = 12 if rng.random() < 0.03 else 7
        c = np.unique(np.clip(r + np.arange(-(k // 2), k - k // 2) * 3, 0, n - 1))
        rows.append(c)
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    nnz = int(rp[-1])
    ok, read, spilled = capi.host_sell_spilled(rp, cols)
    assert ok and 0 < spilled < 0.03 * nnz and read + 4 * spilled < 1.1 * nnz


def test_alternating_short_and_long_rows_are_too_much_padding():
    # rows of 1 and 9 entries alternate: no cap helps (half of the rows are long), the padding to 9 slots
    # costs more than CSR's indices -> the layout does not qualify (the CSR-stream kernel runs)
    n = 4096
    rows = [np.array([r]) if r % 2 else np.unique(np.clip(r + np.arange(-4, 5), 0, n - 1)) for r in range(n)]
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    assert capi.host_sell_check(rp, cols)[0] is False


def test_wide_banded_rows_qualify():
    # 41 diagonals: width 41 -> three 16-byte code words per thread
    n = 1500
    offs = np.arange(-20, 21)
    rows = [np.array([r + o for o in offs if 0 <= r + o < n]) for r in range(n)]
    rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
    cols = np.concatenate(rows).astype(np.int32)
    ok, slots, dict_entries, code_bytes = capi.host_sell_check(rp, cols)
    assert ok and slots == 3 * CHUNK * 41
    # 41 row patterns per chunk (20 truncated at each end + the full one) x 41 offsets fit the table
    assert code_bytes == 3 * 512 and dict_entries % 41 == 0


def test_empty_pattern():
    assert capi.host_sell_check(np.zeros(1, np.int32), np.zeros(0, np.int32)) == (False, 0, 0, 0)


def test_padding_costs_only_the_lines_it_shares_with_entries():
    # rows of 7 entries, except 16 rows per chunk with 12.  When the long rows sit next to each other (one
    # 128-byte line of every plane) the lanes of the other rows stop loading after 7 slots: slots 7..11 cost
    # 16 rows each, nothing is spilled.  Spread one per line, every line would be read to 12 slots -- the
    # tails go to the spill list instead.
    n = 4 * CHUNK

    def pattern(is_long):
        rows = [np.unique(np.clip(r + np.arange(-(k // 2), k - k // 2) * 3, 0, n - 1))
                for r in range(n) for k in [12 if is_long(r) else 7]]
        rp = np.concatenate([[0], np.cumsum([len(c) for c in rows])]).astype(np.int32)
        return rp, np.concatenate(rows).astype(np.int32), np.diff(rp)

    rp, cols, lens = pattern(lambda r: r % CHUNK < 16)
    ok, read, spilled = capi.host_sell_spilled(rp, cols)
    assert ok and spilled == 0
    assert read == sum(16 * lens[c:c + 16].max() + sum(16 * lens[l:l + 16].max() for l in range(c + 16, c + CHUNK, 16))
                       for c in range(0, n, CHUNK))
    assert read < 1.03 * rp[-1]
    rp, cols, lens = pattern(lambda r: 32 * CHUNK // 512 <= r < n - 32 and r % 16 == 0)
    ok, read, spilled = capi.host_sell_spilled(rp, cols)
    assert ok and spilled == int(np.maximum(lens - 7, 0).sum()) and read <= 7 * n
