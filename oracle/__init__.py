"""Test infrastructure: CPU restatement of the reference hot path (see cpu_oracle.py)."""
