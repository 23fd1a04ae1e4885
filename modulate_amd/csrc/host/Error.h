// Error.h -- the reference's status vocabulary (Modulate/Error.h:5-59) for the Linux host mirror.
// Same enumerator names and order, so eError values are interchangeable with the upstream tree.
#pragma once

#include <cstdio>

enum eError : int
{
    eError_NoError = 0,
    eError_FailedToOpenFile,        // 1
    eError_FailedToCreateDirectory, // 2
    eError_UnknownVersionNumber,    // 3  header magic is neither PS3 nor PS4
    eError_ValueOutOfBounds,        // 4
    eError_AlreadyLoaded,           // 5
    eError_InvalidData,             // 6
    eError_NoData,                  // 7
    eError_FailedToCreateFile,      // 8
    eError_FailedToDeleteFile,      // 9
    eError_FailedToCopyFile,        // 10
    eError_InvalidParameter,        // 11
    eError_FailedToWriteData,       // 12
    eError_NumTypes
};

// Text printed for each code: the reference prints "ERROR: <text>" to stdout (Error.h:22-41).
inline const char* ErrorText( eError leError )
{
    switch( leError )
    {
    case eError_NoError:                 return "No Error";
    case eError_FailedToOpenFile:        return "Failed to open file";
    case eError_FailedToCreateDirectory: return "Failed to create directory";
    case eError_UnknownVersionNumber:    return "Unknown version number";
    case eError_ValueOutOfBounds:        return "Value of out bounds";
    case eError_AlreadyLoaded:           return "Already loaded";
    case eError_InvalidData:             return "Bad data";
    case eError_NoData:                  return "Missing data";
    case eError_FailedToCreateFile:      return "Failed to create file";
    case eError_FailedToDeleteFile:      return "Failed to delete file";
    case eError_FailedToCopyFile:        return "Failed to copy file";
    case eError_InvalidParameter:        return "Invalid parameter";
    case eError_FailedToWriteData:       return "Failed to write data";
    default:                             return "Unknown error";
    }
}

inline void ShowError( eError leError ) { std::printf( "ERROR: %s\n", ErrorText( leError ) ); }

#define ERROR_RETURN                     do { if( leError != eError_NoError ) return leError; } while( 0 )
#define SHOW_ERROR_AND_RETURN            do { if( leError != eError_NoError ) { ShowError( leError ); return leError; } } while( 0 )
#define SHOW_ERROR_AND_RETURN_W( lTodo ) do { if( leError != eError_NoError ) { ShowError( leError ); lTodo; return leError; } } while( 0 )
