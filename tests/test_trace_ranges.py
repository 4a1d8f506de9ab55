"""roctx stage ranges: a no-op without the library or with PT_ROCTX=0, never an error (the metric dict keeps the reference's shape either way)."""
import importlib


def test_stage_range_is_safe_without_a_profiler(monkeypatch):
    import pdf_table_amd.trace_ranges as tr
    with tr.stage_range("text_detection"):
        with tr.stage_range("nested"):
            pass
    monkeypatch.setenv("PT_ROCTX", "0")
    tr = importlib.reload(tr)
    assert not tr.available()
    with tr.stage_range("layout"):
        pass
    monkeypatch.delenv("PT_ROCTX")
    importlib.reload(tr)


def test_pipeline_names_the_four_stages():
    import inspect
    from pdf_table_amd import pipeline
    src = inspect.getsource(pipeline.OcrTablePipeline)
    for name in ("layout", "text_detection", "text_recognition", "table_structure"):
        assert f'stage_range("{name}")' in src, name
