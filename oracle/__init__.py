"""CPU oracle (test infrastructure only) -- see oracle/kf_oracle.c header."""
