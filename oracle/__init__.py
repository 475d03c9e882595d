"""Test-infrastructure oracle (see bipgo_oracle.py header).  Never imported by vican_amd."""
