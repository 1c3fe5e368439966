"""CPU oracle: test infrastructure only (tests/, __graft_entry__.smoke(), bench.py's cpu_baseline).  The product path
(scptoolbox.jl_amd) never imports it.  Parity status: "parity unpinned" -- the reference ships no golden vectors and
cannot be run here (no Julia / ECOS); every restatement cites the reference lines it follows and is pinned on
mathematics (tests/test_oracle_*.py)."""
