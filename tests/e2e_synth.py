"""Seeded inputs of the end-to-end fixture shared by tests/golden/make_golden.py::gen_e2e_page (the oracle chain's run) and
tests/test_gpu_e2e.py (the engine's run): page indices, weights, table regions."""
import numpy as np

E2E_PAGES = (17, 6)         # synthetic page indices (pdf_table_amd.synth_pages.make_page): a two-table and a one-table page whose oracle logical locations keep clear of the .5 rounding boundary on two of the three tables, so that HTML strings get compared


def e2e_state_dicts():
    """the seeded weights of the end-to-end fixture = bench.py's timed checkpoint set (pdf_table_amd.synth_weights.conditioned_state_dicts): the
    detector carries the hand-built text channel (boxes come out), the recogniser the fitted classifier (tools/fit_crnn_classifier.py), the Lore
    detector the heat-map bias that yields cells and the small DCN offsets of the full-size parity test, the layout net the head branch fitted to the
    generator's pages (tools/fit_layout_head.py): its "table" regions are what the table stage crops, as in the reference"""
    from pdf_table_amd import synth_weights as sw
    return sw.conditioned_state_dicts()


def e2e_table_boxes(meta):
    t = np.asarray(meta["tables"]).astype(np.int64).reshape(-1, 4)
    return np.stack([np.maximum(t[:, 0] - 8, 0), np.maximum(t[:, 1] - 8, 0), np.minimum(t[:, 2] + 8, 1024), np.minimum(t[:, 3] + 8, 1024)], 1)
