from .data_loaders import YELPDataLoader, build_vocab, chat_tokenize, process_reviews, PAD, UNK, BOS, EOS, SYS, USR  # noqa: F401
