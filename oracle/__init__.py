"""TEST INFRASTRUCTURE ONLY — CPU checker for the ISLE hot path (see isle_oracle.cpp header)."""
