"""Native LETOR parser (csrc/letor.cpp behind ptr_letor_scan / ptr_letor_load) and the host grouping / scaling / filtering
around it, against fixtures produced by the reference's own parse_letor / iter_queries (tests/golden/make_golden_letor.py).
Host code only: runs without a GPU."""
import os

import numpy as np
import pytest

from ptranking_amd import letor
from ptranking_amd.batching import PaddedQueryBatches

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ONE = os.path.join(GOLD, "letor_sample.txt")
ZERO = os.path.join(GOLD, "letor_sample_zero.txt")


@pytest.fixture(scope="module")
def gold():
    return np.load(os.path.join(GOLD, "letor.npz"))


@pytest.mark.parametrize("tag,path,one_indexed", [("one", ONE, True), ("zero", ZERO, False)])
def test_parse_matches_reference_bit_exact(gold, tag, path, one_indexed):
    X64, y, qids, qoff = letor.parse_letor_file(path, one_indexed=one_indexed, dtype=np.float64)
    assert np.array_equal(X64, gold[f"{tag}/X"])                       # strtod == Python float(): identical doubles
    assert np.array_equal(y.astype(np.float64), gold[f"{tag}/y"])
    per_doc = np.repeat(qids, np.diff(qoff))
    assert np.array_equal(per_doc, gold[f"{tag}/qids"])
    X32, y2, _, _ = letor.parse_letor_file(path, one_indexed=one_indexed)
    assert X32.dtype == np.float32
    assert np.array_equal(X32, gold[f"{tag}/X"].astype(np.float32))    # the FloatTensor cast of data_utils.py:610
    assert qoff[0] == 0 and qoff[-1] == X64.shape[0] and np.all(np.diff(qoff) > 0)


# (fixture tag, loader kwargs, tolerance on the float32 features)
ITER_CASES = [
    ("iter_plain", dict(min_docs=1, min_rele=1), 0.0),
    ("iter_filter", dict(min_docs=5, min_rele=1, binary_rele=True), 0.0),
    ("iter_std", dict(min_docs=1, min_rele=1, scaler_id="StandardScaler"), 1e-6),
    ("iter_minmax", dict(min_docs=2, min_rele=1, unknown_as_zero=True, scaler_id="MinMaxScaler"), 1e-6),
    ("iter_robust", dict(min_docs=1, min_rele=1, scaler_id="RobustScaler"), 1e-6),
]


@pytest.mark.parametrize("tag,kw,tol", ITER_CASES, ids=[c[0] for c in ITER_CASES])
def test_queries_match_reference_iter_queries(gold, tag, kw, tol):
    Qs = letor.load_letor_queries(ONE, presort=False, **kw)
    assert len(Qs) == int(gold[f"{tag}/n"])
    for i, (qid, x, y) in enumerate(Qs):
        assert qid == int(gold[f"{tag}/q{i}/qid"])
        ref = gold[f"{tag}/q{i}/X"].astype(np.float32)
        assert x.dtype == np.float32 and x.shape == ref.shape
        if tol == 0.0:
            assert np.array_equal(x, ref)
        else:   # scaling re-associates float64 sums differently from sklearn: 1e-6 absolute + relative on the fp32 result
            assert np.all(np.abs(x - ref) <= tol + tol * np.abs(ref)), np.abs(x - ref).max()
        assert np.array_equal(y.astype(np.float64), gold[f"{tag}/q{i}/y"])


def test_yahoo_zero_indexed(gold):
    Qs = letor.load_letor_queries(ZERO, presort=False, min_docs=1, min_rele=1, one_indexed=False)
    assert len(Qs) == int(gold["iter_yahoo/n"])
    for i, (qid, x, y) in enumerate(Qs):
        assert qid == int(gold[f"iter_yahoo/q{i}/qid"])
        assert np.array_equal(x, gold[f"iter_yahoo/q{i}/X"].astype(np.float32))


def test_split_runs_merge_in_first_seen_order(gold):
    _, _, run_qids, _ = letor.parse_letor_file(ONE)
    assert list(run_qids) == [101, 7, 350, 12, 7, 99, 1000, 64]
    Qs = letor.load_letor_queries(ONE, presort=False)
    assert [q[0] for q in Qs] == [101, 7, 350, 12, 99, 1000, 64]
    assert Qs[1][1].shape[0] == 17          # 12 + 5 rows of qid 7


def test_presort_is_label_descending():
    for _, _, y in letor.load_letor_queries(ONE, presort=True):
        assert np.all(np.diff(y) <= 0)


def test_comments_blank_lines_crlf_and_errors(tmp_path):
    p = tmp_path / "c.txt"
    p.write_text("# header comment\n\n2 qid:5 1:0.5 3:1e-2 # docid = GX000-00 inc = 1\r\n0 qid:5 2:-4\n   \n1 qid:abc 4:7 #x\n")
    X, y, qids, qoff = letor.parse_letor_file(str(p))
    assert X.shape == (3, 4)
    assert np.array_equal(X, np.array([[0.5, 0, 0.01, 0], [0, -4, 0, 0], [0, 0, 0, 7]], np.float32))
    assert list(y) == [2, 0, 1] and list(qoff) == [0, 2, 3] and qids[0] == 5 and qids[1] > 0
    Xm, *_ = letor.parse_letor_file(str(p), missing=-1.0)
    assert Xm[0, 1] == -1.0 and Xm[0, 0] == 0.5
    bad = tmp_path / "bad.txt"
    bad.write_text("1 qid:1 1:0.5\n1 quid:1 1:0.5\n")
    with pytest.raises(ValueError, match="malformed line 2"):
        letor.parse_letor_file(str(bad))
    with pytest.raises(ValueError, match="cannot open"):
        letor.parse_letor_file(str(tmp_path / "missing.txt"))
    empty = tmp_path / "empty.txt"
    empty.write_text("")
    X, y, qids, qoff = letor.parse_letor_file(str(empty))
    assert X.shape[0] == 0 and list(qoff) == [0]


def test_threaded_parse_of_a_larger_file(tmp_path):
    rng = np.random.default_rng(3)
    n_q, F = 400, 40
    X = np.round(rng.standard_normal((n_q * 25, F)), 5)
    lab = rng.integers(0, 5, n_q * 25)
    with open(tmp_path / "big.txt", "w") as f:
        for i in range(X.shape[0]):
            f.write(f"{lab[i]} qid:{i // 25 + 1} " + " ".join(f"{k + 1}:{float(X[i, k])!r}" for k in range(F)) + "\n")
    Xp, y, qids, qoff = letor.parse_letor_file(str(tmp_path / "big.txt"), dtype=np.float64)
    assert np.array_equal(Xp, X) and np.array_equal(y, lab.astype(np.float32))
    assert np.array_equal(qids, np.arange(1, n_q + 1)) and np.array_equal(np.diff(qoff), np.full(n_q, 25))


def test_padded_batches_from_letor_file():
    pb = PaddedQueryBatches.from_letor_file(ONE, "cpu", rough_batch_size=64, pad_to=8, min_docs=2, min_rele=1)
    ids = [q for b in pb for q in b[0]]
    assert sorted(ids) == sorted([101, 7, 12, 1000, 64])      # 350 has 1 document, 99 has no relevant one
    for _, X, Y, lens in pb:
        assert X.shape[:2] == Y.shape and X.shape[2] == 23 and X.shape[1] % 8 == 0
        for r in range(X.shape[0]):
            n = int(lens[r])
            assert np.all(np.diff(Y[r, :n].numpy()) <= 0) and float(Y[r, n:].abs().sum()) == 0.0


def test_value_spellings_round_like_python_float(tmp_path):
    """The parser's short-decimal fast path and its strtod fallback must both give Python float()'s double."""
    rng = np.random.default_rng(11)
    toks = ["0", "-0", "0.0", "-0.000", "1e22", "1e-22", "9.999999999999999e22", "123456789012345e7", "123456789012345e-22",
            "1234567890123456", "0.1", "0.30000000000000004", "5e-324", "1.7976931348623157e308", "1e23", "8.5e-23", ".5", "5.",
            "+3.25", "1E5", "1e+5", "000123.4500", "0.000001234", "4.35", "2.675", "1.005e2", "nan", "inf", "-inf"]
    for _ in range(3000):
        v = rng.standard_normal() * 10.0 ** int(rng.integers(-12, 13))
        toks.append([f"{v:.{int(rng.integers(0, 9))}f}", f"{v:.{int(rng.integers(0, 17))}e}", repr(float(v)),
                     f"{v:.{int(rng.integers(1, 18))}g}"][int(rng.integers(0, 4))])
    with open(tmp_path / "v.txt", "w") as f:
        for i in range(0, len(toks), 50):
            f.write("1 qid:1 " + " ".join(f"{k + 1}:{t}" for k, t in enumerate(toks[i:i + 50])) + "\n")
    X, *_ = letor.parse_letor_file(str(tmp_path / "v.txt"), dtype=np.float64)
    got = X.reshape(-1)[:len(toks)]
    want = np.array([float(t) for t in toks])
    same = (got == want) | (np.isnan(got) & np.isnan(want))
    assert same.all(), [(toks[i], got[i], want[i]) for i in np.flatnonzero(~same)[:5]]
    assert np.array_equal(np.signbit(got), np.signbit(want))


@pytest.mark.parametrize("text,line", [("1 qid:1 1:0.5 2:\n2 qid:2 1:0.25\n", 1),      # "fid:" with no value at the end of a line
                                       ("1 qid:\n2 qid:2 1:0.25\n", 1),                # "qid:" with nothing behind it
                                       ("1 qid:1 1:0.5\n0 qid:2 1:", 2)])              # ... on the last line, no trailing newline
def test_truncated_tokens_are_flagged_not_read_through(tmp_path, text, line):
    """strtod / strtoll skip leading whitespace INCLUDING newlines: a value or query id that is missing at the end of a line must
    not be taken from the next line (or from beyond the buffer on the last one)."""
    from ptranking_amd import letor
    f = tmp_path / "trunc.txt"
    f.write_text(text)
    with pytest.raises(ValueError, match=f"malformed line {line}"):
        letor.parse_letor_file(str(f))
