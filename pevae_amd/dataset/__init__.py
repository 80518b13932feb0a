from .data_loaders import YELPDataLoader, build_vocab, PAD, UNK, BOS, EOS, SYS, USR  # noqa: F401
