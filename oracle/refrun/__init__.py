"""Reference-run harness (THIS CONTAINER ONLY): executes the reference's own Python files from
/root/reference behind small shims to pin the oracle and to generate tests/golden/*.npz.
Never imported on the GPU box (there is no /root/reference there)."""
