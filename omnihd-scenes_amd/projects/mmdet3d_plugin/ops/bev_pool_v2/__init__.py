# BEVPoolv2 operator package (same module path as the reference plugin).
