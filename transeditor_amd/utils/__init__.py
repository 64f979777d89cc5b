"""Host-side helpers mirroring the reference's `utils/` modules used by the train step."""
