"""vidar_amd.plugin.modules -- see vidar_amd/plugin/__init__.py for the registry surface."""
