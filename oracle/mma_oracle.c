/* placeholder translation unit: the MMA restatement lands here. */
typedef int orc_mma_placeholder_t;
