"""Import-path aliases: the reference's `det3d.*` module paths (the `_target_`s of its YAML configs) resolve to the
MI355X-native implementations in `pillarnext_amd`.  Nothing here is reference code."""
