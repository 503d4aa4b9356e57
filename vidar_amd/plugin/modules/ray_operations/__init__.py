"""vidar_amd.plugin.ray_operations -- see vidar_amd/plugin/__init__.py for the registry surface."""
