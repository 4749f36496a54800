"""CPU oracle (test infrastructure only) - see m2fnet_oracle.py."""
