! blk_io.f -- whole-COMMON-block stream I/O: the actual argument is the first member of a block,
! the dummy spans the block (storage association of COMMON members).
      subroutine blk_read(iu, a, n)
      integer iu, n
      double precision a(n)
      read(iu) a
      end
      subroutine blk_write(iu, a, n)
      integer iu, n
      double precision a(n)
      write(iu) a
      end
