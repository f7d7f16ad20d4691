"""`from global_settings import GlobalSettings as gs` for callers written against the reference (modules/global_settings.py:9-81): a class whose
attributes ARE the values of `camera_linearity_amd.settings` (read at access time, so `settings.configure(...)` shows through, and assignable:
`GlobalSettings.DARK_THRESHOLD = 0.02` configures). The reference fills the class from data/config.ini at import; nothing is read from disk
here. Settings of subsystems outside the hot path (DoRF / PCA data files, plotting constants) are not defined: AttributeError names them."""
from . import settings as _settings


class _Meta(type):
    def __getattr__(cls, name):
        if name.isupper() and hasattr(_settings, name):
            return getattr(_settings, name)
        raise AttributeError(f"GlobalSettings.{name} is not a setting of the merge / linearity / calibration-energy path "
                             f"(defined: {', '.join(sorted(k for k in vars(_settings) if k.isupper()))})")

    def __setattr__(cls, name, value):
        if name.isupper():
            _settings.configure(**{name: value})
        else:
            super().__setattr__(name, value)

    def __dir__(cls):
        return sorted(set(super().__dir__()) | {k for k in vars(_settings) if k.isupper()})


class GlobalSettings(metaclass=_Meta):
    """modules/global_settings.py:9 - class-level settings, backed by camera_linearity_amd.settings."""
