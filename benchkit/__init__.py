"""The pieces of bench.py (the driver-facing entry point at the repo root): workloads and constants, the ranks of an
N > 1 run, the timed measurement, the record (roofline, comm block), the parity checks against the oracle, the extra
workloads / divisions of one record.  Measurement infrastructure: nothing here is imported by clane_amd/."""
