"""oracle — TEST INFRASTRUCTURE, NOT PRODUCT.

CPU restatements of the reference algorithms on the hot path.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this package; the
product (``omnihd-scenes_amd/``) never does.
"""
