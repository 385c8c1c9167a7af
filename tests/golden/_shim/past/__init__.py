"""Minimal stand-in for the `future` package's `past` module (not installed in
this image; no network).  Used ONLY by tests/golden/make_golden.py, in the
build container, to import the read-only reference at /root/reference and
record golden vectors.  Never imported by the product, the tests or the bench."""
