"""Host-side mirrors of the reference's hot-path nn.Module classes (same constructor / forward
signatures and state_dict keys); arithmetic runs in libcvc_hip.so."""
