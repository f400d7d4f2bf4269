"""Test infrastructure only: CPU restatement of the reference's hot path.  See lanegcn_oracle.py."""
