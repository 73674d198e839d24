"""Import-path shim: `posterior_matching.*` as in the reference, backed by posterior_matching_amd."""
