"""oracle/ -- CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.

Nothing under ``pdf_table_amd/`` (the product) may import this package.  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it, and only as the
checker / the thing timed beside the GPU, never as a fallback.

Every function cites the reference file:line it restates (paths relative to
``/root/reference/src/pdftable``).  Pinning status (see DESIGN.md "Oracle"):

* float nets (DB-ResNet18, CRNN)           -- PINNED: golden vectors in ``tests/golden/*.npz`` were
  produced by importing the reference ``nn.Module`` files themselves (``tests/golden/make_golden.py``).
* CTC decode, order_point, box ordering     -- PINNED the same way (pure numpy/torch reference code).
* cv2 / pyclipper / shapely arithmetic (resize, findContours, minAreaRect, fillPoly, polygon
  offsetting, warpPerspective)              -- PARITY UNPINNED: those libraries are not installed in
  the build image and the reference holds no test vectors for them; the restatements follow the
  libraries' published algorithms and are pinned only by hand-derived known answers.
"""
