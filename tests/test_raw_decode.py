"""mgx_results_from_raw on untrusted input (it decodes records that arrived over the wire from other ranks): a well-formed
record set decodes; a stream cut short, a header pointing past the stream and a CIGAR word whose operator is not one of
the six the formatter knows (Cigar::Operator, graph/alignment/aligner_cigar.hpp:18-25) are refused with
MGX_ERR instead of being indexed with."""
import numpy as np
import pytest

from metagraph_amd import capi, gather as mg
import emu_drv
from test_emu_vs_oracle import make_world


@pytest.fixture(scope="module")
def raw_records():
    k = 15
    g, reads = make_world(5, k, genome_len=2000, n_reads=12, read_len=80, n_variants=6)
    run = emu_drv.EmuRun(emu_drv.EmuGraph(g), capi.config_cli(k), reads)
    assert not run.error
    hb, sb = run.raw()
    return np.frombuffer(hb, dtype=np.uint8).copy(), np.frombuffer(sb, dtype=np.uint8).copy()


def _first_aligned(h):
    """-> (stream_off, n_nodes, n_cigar) of the first read with an alignment (ReadResult, align_types.hpp:41-54)"""
    rec = h.reshape(-1, 64)
    for i in range(rec.shape[0]):
        w = rec[i, :32].view(np.int32)
        if w[0] == 0 and w[1] > 0:
            return int(rec[i, 32:40].view(np.uint64)[0]), int(w[4]), int(w[5])
    raise AssertionError("no aligned read")


def test_well_formed_decodes(raw_records):
    h, s = raw_records
    raw = mg.RawResults(h, s)
    assert raw.res.n_queries == h.size // 64
    raw.close()


def test_truncated_stream_is_refused(raw_records):
    h, s = raw_records
    with pytest.raises(RuntimeError):
        mg.RawResults(h, s[: (s.size // 8) * 4])


def test_bad_cigar_operator_is_refused(raw_records):
    h, s = raw_records
    stream_off, n_nodes, n_cigar = _first_aligned(h)
    assert n_cigar > 0
    words = s.view(np.uint32)
    for bad in (6, 7):
        w2 = words.copy()
        w2[stream_off + n_nodes] = (w2[stream_off + n_nodes] & ~np.uint32(7)) | np.uint32(bad)
        with pytest.raises(RuntimeError):
            mg.RawResults(h, w2.view(np.uint8))


def test_header_pointing_past_the_stream_is_refused(raw_records):
    h, s = raw_records
    h2 = h.copy()
    rec = h2.reshape(-1, 64)
    for i in range(rec.shape[0]):
        w = rec[i, :32].view(np.int32)
        if w[0] == 0 and w[1] > 0:
            rec[i, 32:40].view(np.uint64)[0] = s.size          # in words: 4 x past the end
            break
    with pytest.raises(RuntimeError):
        mg.RawResults(h2, s)


def test_labeled_records_decode_with_their_label_lists():
    """records of a label-aware aligner (every alignment followed by its label list in the stream) through
    mgx_results_from_raw_labeled: the same alignments and labels as the run's own decode; a label count pointing past the
    stream is refused"""
    import orc
    from labeled_worlds import labeled_world
    g, anno, reads = labeled_world(3, 11, n_strains=3, n_reads=16)
    run = emu_drv.EmuRun(emu_drv.EmuGraph(g), capi.config_cli(11), reads, annotation=emu_drv.EmuAnnotation(anno))
    assert not run.error
    hb, sb = run.raw()
    h, s = np.frombuffer(hb, dtype=np.uint8).copy(), np.frombuffer(sb, dtype=np.uint8).copy()
    raw = mg.RawResults(h, s, labeled=True)
    got = capi.results_to_py(raw.res)
    want, _ = run.results()
    assert got == want and any("labels" in a and a["labels"] for q in got for a in q)
    raw.close()
    with pytest.raises(RuntimeError):
        mg.RawResults(h, s[: s.size - 4], labeled=True)          # the last alignment's labels are cut off
