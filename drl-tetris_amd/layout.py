"""Word indices of the device state layout (mirror of csrc/tetris_layout.h) for host-side code that
edits snapshot blobs."""
NCOL = 10
W_COL0, W_PIECE, W_MISC, W_TIME = 0, 10, 11, 12
NWORDS = 39
NGWORDS = 5
G_META, G_EPISODE, G_STEPS, G_LINES, G_SENT = 0, 1, 2, 3, 4
