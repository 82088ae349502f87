"""Module paths of the reference's ``utils`` package, kept so that its scripts and pickles (which name
``utils.coma.negative_exp``) resolve here.  Implementations live in ``coma_amd``."""
