﻿!mod$ v1 sum:dc0e186b83d0ae94
!need$ fadd42cafe0c8e6b n m_io_session
module m_checkpoint_state
use m_io_session,only:reader_session_t
use m_io_session,only:writer_session_t
type,abstract::checkpoint_state_t
contains
procedure(write_checkpoint_iface),deferred::write_checkpoint
procedure(read_checkpoint_iface),deferred::read_checkpoint
end type
abstract interface
subroutine write_checkpoint_iface(self,writer)
import::checkpoint_state_t
import::writer_session_t
class(checkpoint_state_t),intent(inout)::self
type(writer_session_t),intent(inout)::writer
end
end interface
abstract interface
subroutine read_checkpoint_iface(self,reader)
import::checkpoint_state_t
import::reader_session_t
class(checkpoint_state_t),intent(inout)::self
type(reader_session_t),intent(inout)::reader
end
end interface
end
