"""Drop-in mirror of the reference's ``icv_src`` package for the hot path only
(``icv_encoder``, ``icv_model``, ``icv_module``): same import paths, class names, constructor
arguments and error behaviour, backed by the MI355X-native engine in ``licv``."""
