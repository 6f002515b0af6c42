"""Word indices of the device state layout (mirror of csrc/tetris_layout.h) for host-side code that
reads or edits snapshot blobs."""
NCOL = 10
W_COL0, W_PIECE, W_MISC, W_TIME, W_DROPCOMBO = 0, 10, 11, 12, 13
W_PIECE_DRAWS, W_HOLE_DRAWS = 21, 22
NWORDS = 39
NGWORDS = 5
G_META, G_EPISODE, G_STEPS, G_LINES, G_SENT = 0, 1, 2, 3, 4
