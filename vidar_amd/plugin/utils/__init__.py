"""vidar_amd.plugin.utils -- see vidar_amd/plugin/__init__.py for the registry surface."""
