"""Seeded synthetic workloads (clouds A / B, orbit cameras, upstream gradients, the hybrid tracer's mesh scenes, the
training surrogate) shared by bench.py, __graft_entry__.smoke(), the golden generators and the tests.  Not part of the product
package `3dgrut_amd/` (which holds only the plugins and the C-ABI library) and not test code either: bench.py must not import tests/."""
