"""TEST INFRASTRUCTURE ONLY — CPU restatement of the reference's spectrum path.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package; the product (sdr_iq_visualizer_amd) never does.
"""
