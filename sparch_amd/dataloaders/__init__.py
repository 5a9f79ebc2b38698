"""Data loaders mirroring the reference's `sparch.dataloaders` for the path this build covers (SURVEY f-3)."""
