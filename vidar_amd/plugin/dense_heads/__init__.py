"""vidar_amd.plugin.dense_heads -- see vidar_amd/plugin/__init__.py for the registry surface."""
