// CDtaFile.cpp -- see CDtaFile.h.  Bounds-checked where the reference walks raw pointers.
#include "CDtaFile.h"
#include "Settings.h"

#include <cstdio>
#include <cstring>
#include <sstream>

namespace
{
struct Reader
{
    const unsigned char* p;
    size_t n, at;
    bool Has( size_t k ) const { return n - at >= k; }
    template < typename T > bool Get( T& v )
    {
        if( !Has( sizeof( T ) ) ) return false;
        std::memcpy( &v, p + at, sizeof( T ) );
        at += sizeof( T );
        return true;
    }
};

template < typename T > void Put( std::vector< unsigned char >& o, T v )
{
    unsigned char b[ sizeof( T ) ];
    std::memcpy( b, &v, sizeof( T ) );
    o.insert( o.end(), b, b + sizeof( T ) );
}

constexpr int kiMaxDepth = 256;

// CDtaFile::AddTreeNode (CDtaFile.cpp:393-509)
eError ReadTreeBody( Reader& r, SDtaNode& lNode, int liDepth )
{
    if( liDepth > kiMaxDepth ) return eError_InvalidData;
    int16_t lsNumChildren = 0, lsNodeId = 0;
    if( !r.Get( lsNumChildren ) || lsNumChildren <= 0 ) return eError_InvalidData; // CDtaFile.cpp:398-401
    if( !r.Get( lsNodeId ) ) return eError_InvalidData;
    lNode.msNodeId = lsNodeId;
    lNode.maChildren.reserve( (size_t)lsNumChildren );
    for( int ii = 0; ii < lsNumChildren; ++ii )
    {
        SDtaNode lChild;
        int32_t liType = 0;
        if( !r.Get( liType ) ) return eError_InvalidData;
        lChild.miType = liType;
        switch( liType )
        {
        case ENodeType_String:
        case ENodeType_IncludeFile:
        case ENodeType_Define:
        case ENodeType_Id:
        {
            int32_t liLength = 0;
            if( !r.Get( liLength ) || liLength < 0 || !r.Has( (size_t)liLength ) ) return eError_InvalidData;
            lChild.mString.assign( reinterpret_cast< const char* >( r.p + r.at ), (size_t)liLength );
            size_t z = lChild.mString.find( '\0' ); // the reference goes through a C string (CDtaFile.cpp:448-455)
            if( z != std::string::npos ) lChild.mString.resize( z );
            r.at += (size_t)liLength;
            break;
        }
        case ENodeType_Tree1:
        case ENodeType_Tree2:
        {
            int32_t liSkipped = 0; // written as 1 (CDtaFile.cpp:1319), ignored on load (:467)
            if( !r.Get( liSkipped ) ) return eError_InvalidData;
            eError leError = ReadTreeBody( r, lChild, liDepth + 1 );
            if( leError != eError_NoError ) return leError;
            break;
        }
        case ENodeType_Integer0:
        case ENodeType_Integer6:
        case ENodeType_Integer8:
        case ENodeType_Integer9:
            if( !r.Get( lChild.miValue ) ) return eError_InvalidData;
            break;
        case ENodeType_Float:
            if( !r.Get( lChild.mfValue ) ) return eError_InvalidData;
            break;
        default:
            return eError_InvalidData; // CDtaFile.cpp:503-504
        }
        lNode.maChildren.push_back( std::move( lChild ) );
    }
    return eError_NoError;
}

// CDtaNodeBase::SaveToStream + the leaf writers (CDtaFile.cpp:1310-1326, CDtaFile.h:262-284)
void WriteTreeBody( std::vector< unsigned char >& o, const SDtaNode& lNode )
{
    Put< int16_t >( o, (int16_t)lNode.maChildren.size() );
    Put< int16_t >( o, lNode.msNodeId );
    for( const SDtaNode& c : lNode.maChildren )
    {
        Put< int32_t >( o, c.miType );
        if( c.IsTree() )
        {
            Put< int32_t >( o, 1 );
            WriteTreeBody( o, c );
        }
        else if( c.miType == ENodeType_Float ) Put< float >( o, c.mfValue );
        else if( c.miType == ENodeType_String || c.miType == ENodeType_IncludeFile || c.miType == ENodeType_Define || c.miType == ENodeType_Id )
        {
            Put< int32_t >( o, (int32_t)c.mString.size() );
            o.insert( o.end(), c.mString.begin(), c.mString.end() );
        }
        else Put< int32_t >( o, c.miValue );
    }
}

void DumpNode( std::ostringstream& o, const SDtaNode& n, int liDepth )
{
    o << std::string( (size_t)liDepth * 2, ' ' ) << n.miType;
    if( n.IsTree() )
    {
        o << " tree id=" << n.msNodeId << " n=" << n.maChildren.size() << "\n";
        for( const SDtaNode& c : n.maChildren ) DumpNode( o, c, liDepth + 1 );
    }
    else if( n.miType == ENodeType_Float )
    {
        uint32_t u;
        std::memcpy( &u, &n.mfValue, 4 );
        o << " f32bits=" << u << "\n";
    }
    else if( n.miType == ENodeType_String || n.miType == ENodeType_IncludeFile || n.miType == ENodeType_Define || n.miType == ENodeType_Id )
        o << " str=" << n.mString << "\n";
    else o << " int=" << n.miValue << "\n";
}
} // namespace

eError CDtaFile::LoadFromMemory( const unsigned char* lpData, size_t liSize ) // CDtaFile.cpp:57-100
{
    maTopLevel.clear();
    if( liSize < 5 ) return eError_InvalidData;
    Reader r{ lpData, liSize, 5 }; // the 5-byte prefix (u8 1, i32 1) is skipped unread (CDtaFile.cpp:75)
    int32_t liType = ENodeType_Tree1;
    while( r.at < liSize )
    {
        if( liType != ENodeType_Tree1 && liType != ENodeType_Tree2 ) return eError_InvalidData;
        SDtaNode lNode;
        lNode.miType = liType;
        eError leError = ReadTreeBody( r, lNode, 0 );
        if( leError != eError_NoError ) return leError;
        maTopLevel.push_back( std::move( lNode ) );
        if( r.at >= liSize ) break;
        int32_t liOne = 0;
        if( !r.Get( liType ) || !r.Get( liOne ) ) return eError_InvalidData; // next node's type + the int after it (:95-96)
    }
    return eError_NoError;
}

eError CDtaFile::Load( const char* lpFilename )
{
    FILE* f = lpFilename ? std::fopen( lpFilename, "rb" ) : nullptr;
    if( !f )
    {
        eError leError = eError_FailedToOpenFile;
        SHOW_ERROR_AND_RETURN;
    }
    std::fseek( f, 0, SEEK_END );
    long n = std::ftell( f );
    std::fseek( f, 0, SEEK_SET );
    std::vector< unsigned char > lData( n > 0 ? (size_t)n : 0 );
    size_t got = lData.empty() ? 0 : std::fread( lData.data(), 1, lData.size(), f );
    std::fclose( f );
    eError leError = got == lData.size() ? LoadFromMemory( lData.data(), lData.size() ) : eError_InvalidData;
    SHOW_ERROR_AND_RETURN;
    return eError_NoError;
}

void CDtaFile::SaveToMemory( std::vector< unsigned char >& lOut ) const // CDtaFile.cpp:362-391
{
    lOut.clear();
    lOut.push_back( 1 );
    Put< int32_t >( lOut, 1 );
    bool lbFirst = true;
    for( const SDtaNode& n : maTopLevel )
    {
        // The reference writes top-level trees back to back (CDtaFile.cpp:371-374) -- the default here
        // too, although its own Load cannot read that back when there are several; -fixquirks writes
        // the separator Load expects between them (CDtaFile.cpp:95-96).
        if( !lbFirst && CSettings::mbFixReferenceQuirks )
        {
            Put< int32_t >( lOut, n.miType );
            Put< int32_t >( lOut, 1 );
        }
        WriteTreeBody( lOut, n );
        lbFirst = false;
    }
}

eError CDtaFile::Save( const char* lpFilename ) const
{
    std::vector< unsigned char > lOut;
    SaveToMemory( lOut );
    FILE* f = lpFilename ? std::fopen( lpFilename, "wb" ) : nullptr;
    if( !f )
    {
        eError leError = eError_FailedToCreateFile;
        SHOW_ERROR_AND_RETURN;
    }
    size_t w = std::fwrite( lOut.data(), 1, lOut.size(), f );
    std::fclose( f );
    return w == lOut.size() ? eError_NoError : eError_FailedToWriteData;
}

std::string CDtaFile::Dump() const
{
    std::ostringstream o;
    for( const SDtaNode& n : maTopLevel ) DumpNode( o, n, 0 );
    return o.str();
}
