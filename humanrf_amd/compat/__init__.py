"""Shims with the call signatures of the third-party packages the reference's hot path imports."""
