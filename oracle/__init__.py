"""TEST INFRASTRUCTURE ONLY.  CPU oracle for the Open-Sora denoise path (see the module headers).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package."""
